import Vector3 from './Vector3.js';
import Material from './Material.js';

/**
 * Sphere — the only primitive (raytracer-js-master/src/Sphere.js:7-84).  Public mutable fields
 * center / radius / radius2 / material; the host re-reads them on every render() call, so
 * assigning `sphere.center = new Vector3(..)` between frames works as in the reference's examples.
 */
export default class Sphere {
    constructor(center, radius, material) {
        this.center = center;
        this.radius = radius;
        this.radius2 = radius * radius;
        this.material = material;
    }

    /**
     * Geometric ray/sphere test for user code (the render path runs its own copy on the GPU).
     * Writes out.t0 / out.t1 and returns true on a hit; a sphere whose centre lies behind the ray
     * (tca < 0) is a miss even when the origin is inside it, exactly as in the reference.
     */
    intersect(rayOrigin, rayDir, out) {
        const lx = this.center.x - rayOrigin.x, ly = this.center.y - rayOrigin.y, lz = this.center.z - rayOrigin.z;
        const tca = lx * rayDir.x + ly * rayDir.y + lz * rayDir.z;
        if (tca < 0) return false;
        const d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
        if (d2 > this.radius2) return false;
        const thc = Math.sqrt(this.radius2 - d2);
        out.t0 = tca - thc;
        out.t1 = tca + thc;
        return true;
    }

    getCenter() { return this.center; }
    getRadius() { return this.radius; }
    getMaterial() { return this.material; }

    getNormal(point) { return point.clone().subtract(this.center).normalize(); }

    /** wire format shared with the reference's workers (Sphere.js:53-69) */
    serialize() {
        const m = this.material, sc = m.surfaceColor, ec = m.emissionColor;
        return {
            center: [this.center.x, this.center.y, this.center.z],
            radius: this.radius,
            material: {
                surfaceColor: [sc.x, sc.y, sc.z],
                emissionColor: [ec.x, ec.y, ec.z],
                transparency: m.transparency,
                reflection: m.reflection,
            },
        };
    }

    static deserialize(data) {
        const m = data.material;
        return new Sphere(new Vector3(...data.center), data.radius,
            new Material(new Vector3(...m.surfaceColor), m.reflection, m.transparency, new Vector3(...m.emissionColor)));
    }
}
