// Shared pieces of the Node examples: the example scene of raytracer-js-master/examples/example1.js:9-45 (the same seven
// spheres and background in example2-4), the animation of example2.js:98-104, and what replaces the browser's canvas
// here: a PPM writer and a hash of the frame.
import fs from 'fs';
import crypto from 'crypto';
import { Vector3, Material, Sphere, Scene } from '../index.js';

const v = (x, y, z) => new Vector3(x, y, z);

/** -> { scene, backgroundColor, moving: { light1, sphere1, sphere2, sphere3, sphere4 } } */
export function exampleScene() {
    const matte = (r, g, b, refl, transp) => new Material(v(r, g, b), refl, transp, v(0, 0, 0));
    const lamp = (e) => new Material(v(0, 0, 0), 0, 0, v(e, e, e));
    const floor = new Sphere(v(0.0, -10004, -20), 10000, matte(0.2, 0.2, 0.2, 0, 0));
    const sphere1 = new Sphere(v(0, 0, -20), 4, matte(1.00, 0.32, 0.36, 1, 0.5));
    const sphere2 = new Sphere(v(5, -1, -15), 2, matte(0.9, 0.76, 0.46, 1, 0));
    const sphere3 = new Sphere(v(5, 0, -25), 3, matte(0.65, 0.77, 0.97, 1, 0));
    const sphere4 = new Sphere(v(-5.5, 0, -15), 3, matte(0.9, 0.9, 0.9, 1, 0));
    const light1 = new Sphere(v(0, 20, -30), 3, lamp(1.2));
    const light2 = new Sphere(v(0, 10, 10), 3, lamp(1));
    const scene = new Scene();
    for (const el of [floor, sphere1, sphere2, sphere3, sphere4, light1, light2]) scene.add(el);     // the order matters (ties, light order)
    return { scene, backgroundColor: v(2.0, 2.0, 2.0), moving: { light1, sphere1, sphere2, sphere3, sphere4 } };
}

/** the scene at clock value t (milliseconds): new centres are ASSIGNED, as the reference's animation loops do */
export function animate(moving, t) {
    moving.light1.center = v(10 * Math.sin(t / 2000), 10, -30);
    moving.sphere1.center = v(0, 5 * Math.sin(t / 1000), -20);
    moving.sphere2.center = v(5, -1 * Math.sin(t / 500), -15);
    moving.sphere3.center = v(5, 6 * Math.cos(t / 1000), -25);
    moving.sphere4.center = v(-5.5, 3 * Math.cos(t / 1000), -15);
}

export function sha256(buffer) {
    return crypto.createHash('sha256').update(Buffer.from(buffer.buffer || buffer, buffer.byteOffset || 0, buffer.byteLength)).digest('hex');
}

/** RGBA8 rows (what RayTracer.render returns, what a canvas ImageData holds) -> binary PPM */
export function writePPM(file, rgba, width, height) {
    const src = new Uint8Array(rgba.buffer || rgba, rgba.byteOffset || 0, width * height * 4);
    const out = Buffer.alloc(width * height * 3);
    for (let i = 0, o = 0; i < width * height * 4; i += 4) { out[o++] = src[i]; out[o++] = src[i + 1]; out[o++] = src[i + 2]; }
    fs.writeFileSync(file, Buffer.concat([Buffer.from(`P6\n${width} ${height}\n255\n`), out]));
}

/**
 * The four matrices a THREE.PerspectiveCamera(fov, aspect, near, far) at `from` looking at `to` (up = +y) holds, as
 * column-major arrays: { from, matrixWorld, matrixWorldInverse, projectionMatrix, projectionMatrixInverse } for RcCamera.setView.
 */
export function perspectiveView(from, to, fov, aspect, near = 0.1, far = 10000) {
    const norm = (a) => { const l = Math.hypot(a[0], a[1], a[2]) || 1; return [a[0] / l, a[1] / l, a[2] / l]; };
    const cross = (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]];
    const z = norm([from[0] - to[0], from[1] - to[1], from[2] - to[2]]);
    const x = norm(cross([0, 1, 0], z)), y = cross(z, x);
    const world = Float64Array.of(x[0], x[1], x[2], 0, y[0], y[1], y[2], 0, z[0], z[1], z[2], 0, from[0], from[1], from[2], 1);
    const dot = (a, b) => a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    const inverse = Float64Array.of(x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0, -dot(x, from), -dot(y, from), -dot(z, from), 1);
    const top = near * Math.tan(Math.PI / 180 * 0.5 * fov), right = top * aspect;
    const a = near / right, b = near / top, c = -(far + near) / (far - near), e = -2 * far * near / (far - near);
    const proj = Float64Array.of(a, 0, 0, 0, 0, b, 0, 0, 0, 0, c, -1, 0, 0, e, 0);
    const projInv = Float64Array.of(1 / a, 0, 0, 0, 0, 1 / b, 0, 0, 0, 0, 0, 1 / e, 0, 0, -1, c / e);
    return { from, matrixWorld: world, matrixWorldInverse: inverse, projectionMatrix: proj, projectionMatrixInverse: projInv };
}

/** Math.random with a fixed seed (the render cache's first frame draws its positions from it): runs become reproducible */
export function seedRandom(seed) {
    let a = seed >>> 0;
    Math.random = () => {                                      // mulberry32
        a = (a + 0x6D2B79F5) >>> 0;
        let t = Math.imul(a ^ (a >>> 15), a | 1);
        t ^= t + Math.imul(t ^ (t >>> 7), t | 61);
        return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
    };
}

/** `--name value` and bare positional arguments of process.argv */
export function args(defaults) {
    const out = { ...defaults, _: [] };
    const a = process.argv.slice(2);
    for (let i = 0; i < a.length; i++) {
        if (a[i].startsWith('--')) {
            const key = a[i].slice(2);
            if (typeof defaults[key] === 'boolean') out[key] = true;
            else out[key] = typeof defaults[key] === 'number' ? Number(a[++i]) : a[++i];
        } else out._.push(a[i]);
    }
    return out;
}
