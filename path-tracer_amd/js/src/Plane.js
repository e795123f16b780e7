import Vector3 from './Vector3.js';
import Material from './Material.js';

/**
 * Plane — an EXTENSION primitive (north_star names a ray-plane test; raytracer-js has none: its only primitive is
 * src/Sphere.js:15-33).  Same shape as Sphere: public mutable fields, intersect(rayOrigin, rayDir, out), getNormal,
 * serialize / deserialize.  The specification of how planes take part in rendering is in include/ptr.h ("planes and
 * Phong"); planes are elements AFTER all spheres whatever their position in the Scene.
 */
export default class Plane {
    constructor(point, normal, material) {
        this.point = point;
        this.normal = normal.clone().normalize();       // the specification works with a unit normal
        this.material = material;
    }
    getPoint() { return this.point; }
    getMaterial() { return this.material; }
    getNormal() { return this.normal.clone(); }
    /** denom = d.n; 0: miss; t = ((p - o).n) / denom; t < 0: miss; else out.t0 = out.t1 = t */
    intersect(rayOrigin, rayDir, out) {
        const n = this.normal, p = this.point;
        const denom = rayDir.x * n.x + rayDir.y * n.y + rayDir.z * n.z;
        if (denom == 0) return false;
        const t = ((p.x - rayOrigin.x) * n.x + (p.y - rayOrigin.y) * n.y + (p.z - rayOrigin.z) * n.z) / denom;
        if (t < 0) return false;
        out.t0 = t; out.t1 = t;
        return true;
    }
    serialize() {
        const m = this.material, v = (a) => [a.x, a.y, a.z];
        return { plane: true, point: v(this.point), normal: v(this.normal),
                 material: { surfaceColor: v(m.surfaceColor), emissionColor: v(m.emissionColor), transparency: m.transparency,
                             reflection: m.reflection, specular: m.specular || 0, shininess: m.shininess || 1 } };
    }
    static deserialize(o) {
        const m = o.material;
        return new Plane(new Vector3(...o.point), new Vector3(...o.normal),
                         new Material(new Vector3(...m.surfaceColor), m.reflection, m.transparency, new Vector3(...m.emissionColor),
                                      m.specular, m.shininess));
    }
}
