import Vector3 from './Vector3.js';

/**
 * Camera(from, to, fov, width, height) — accepted and stored like the reference's
 * (raytracer-js-master/src/Camera.js:5-23).  The reference's render loop ignores it (fixed origin 0,
 * fov 30, looking down -Z; SURVEY.md F4), so RayTracer.render() ignores it too.  The extension
 * entry RayTracer.renderFrame({useCamera: true}) uses basis() to move the GPU camera.
 */
export default class Camera {
    constructor(from, to, fov, width, height) {
        this.fov = fov;
        this.width = width;
        this.height = height;
        this.aspectRatio = width / height;
        this.angle = Math.tan(Math.PI * 0.5 * fov / 180);
        this.from = from;
        this.to = to;
        // unlike the reference (which subtracts in place and so overwrites `to`), keep `to` intact
        this.dir = to.clone().subtract(from).normalize();
    }

    /** orthonormal basis {origin, right, up, forward, tanHalfFov} for ptr_set_camera */
    basis() {
        const f = this.dir;
        let up = new Vector3(0, 1, 0);
        if (Math.abs(f.y) > 0.999) up = new Vector3(0, 0, f.y > 0 ? 1 : -1);
        const right = new Vector3(f.y * up.z - f.z * up.y, f.z * up.x - f.x * up.z, f.x * up.y - f.y * up.x).normalize();
        const u = new Vector3(right.y * f.z - right.z * f.y, right.z * f.x - right.x * f.z, right.x * f.y - right.y * f.x);
        return { origin: this.from, right, up: u, forward: f, tanHalfFov: this.angle };
    }
}
