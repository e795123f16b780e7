import Vector3 from './Vector3.js';
import Color from './Color.js';
import native from './native.js';
import Plane from './Plane.js';

/**
 * RayTracer — drop-in for raytracer-js-master/src/RayTracer.js:11-199 whose per-pixel loop runs on
 * an MI355X.  Same constructor and methods:
 *
 *   new RayTracer(backgroundColor, scene[, camera])
 *   .render(width, height, startY?, scanHeight?, isFirstFrame?) -> ArrayBuffer   (RGBA8)
 *   .trace(rayOrigin, rayDir, depth) -> Vector3                                  (unclamped colour)
 *   .requestPixel(rayDir) -> Color {r, g, b}
 *
 * plus renderAsync() (Promise form, for RenderPlanner), renderFrame() (all rows at their true
 * position), renderInto() (the frame into a buffer of the caller's) and traceBatch().  The scene is re-read from the Scene/Sphere/Material objects on
 * every call (the reference's examples mutate `sphere.center` between frames); nothing is
 * cached on add().  The user's backgroundColor object is never modified (the reference clamps it
 * in place as a side effect of returning it from trace(); do not rely on that).
 *
 * options: { device: 0, mode: 'hybrid' | 'tolerant' | 'exact' | 'f32' }
 *   hybrid (default): float32 kernel + float64 re-trace of uncertain pixels; bytes equal the reference
 *   tolerant: the same with only the discontinuous doubts re-traced; every channel within 1/255 of the reference
 *   exact: every pixel in float64, operation for operation
 *   f32: float32 only (max per-channel error 1/255 except on silhouettes / shadow edges)
 *
 * Extension modes (NOT reference behaviour: the reference stores Material.reflection/.transparency
 * and MAX_RAY_DEPTH but never reads them): renderFrame(w, h, {maxDepth, spp, seed}) follows
 * reflection/refraction rays to maxDepth (<= 16) and averages spp Philox-jittered samples per pixel.
 * Their specification is in include/ptr.h; the default mode and 'exact' render it in float64 bit for bit,
 * 'f32' renders it in float32 (the throughput form).  Scenes may also hold Plane elements and Phong
 * materials (extensions too).  render() itself always has the reference semantics.
 */
export default class RayTracer {
    constructor(backgroundColor, scene, camera, options = {}) {
        this.backgroundColor = backgroundColor;
        this.scene = scene;
        this.camera = camera;
        this._native = native();
        this._device = options.device | 0;
        this._handle = this._native.create(this._device);
        const modes = { hybrid: 0, tolerant: this._native.MODE_TOLERANT, exact: this._native.MODE_EXACT_F64, f32: this._native.MODE_F32_ONLY };
        const mode = options.mode === undefined ? 'hybrid' : options.mode;
        if (!(mode in modes)) throw new TypeError(`unknown mode ${mode}`);
        this._modeFlags = modes[mode];
    }

    /** release the GPU context now (otherwise the garbage collector does) */
    dispose() {
        if (this._handle) { this._native.destroy(this._handle); this._handle = null; }
    }

    // ---- scene upload: 12 planes of n numbers, cx cy cz radius sr sg sb refl transp er eg eb
    static flatten(elements) {
        const n = elements.length;
        const soa = new Float64Array(12 * n);
        for (let i = 0; i < n; i++) {
            const el = elements[i], c = el.center, m = el.material, sc = m.surfaceColor, ec = m.emissionColor;
            soa[i] = c.x; soa[n + i] = c.y; soa[2 * n + i] = c.z;
            soa[3 * n + i] = el.radius;
            soa[4 * n + i] = sc.x; soa[5 * n + i] = sc.y; soa[6 * n + i] = sc.z;
            soa[7 * n + i] = m.reflection; soa[8 * n + i] = m.transparency;
            soa[9 * n + i] = ec.x; soa[10 * n + i] = ec.y; soa[11 * n + i] = ec.z;
        }
        return soa;
    }

    // extension data: planes as 14 rows (px py pz nx ny nz sr sg sb refl transp er eg eb), Phong terms as ks[] then exponent[]
    static flattenPlanes(planes) {
        const n = planes.length, soa = new Float64Array(14 * n);
        planes.forEach((pl, i) => {
            const p = pl.point, nr = pl.normal, m = pl.material, sc = m.surfaceColor, ec = m.emissionColor;
            [p.x, p.y, p.z, nr.x, nr.y, nr.z, sc.x, sc.y, sc.z, m.reflection, m.transparency, ec.x, ec.y, ec.z]
                .forEach((v, row) => { soa[row * n + i] = v; });
        });
        return soa;
    }

    _upload(useCamera) {
        if (!this._handle) throw new Error('RayTracer was disposed');
        const all = this.scene.getElements();
        const planes = all.filter(el => el instanceof Plane);
        const elements = planes.length ? all.filter(el => !(el instanceof Plane)) : all;     // spheres first, then planes
        this._native.setScene(this._handle, RayTracer.flatten(elements), elements.length, 1);
        const ordered = planes.length ? elements.concat(planes) : elements;
        const phong = ordered.some(el => el.material.specular > 0);
        if (planes.length || phong || this._hasExtension) {
            let ph = null;
            if (phong) {
                ph = new Float64Array(2 * ordered.length);
                ordered.forEach((el, i) => { ph[i] = el.material.specular || 0; ph[ordered.length + i] = el.material.shininess || 1; });
            }
            this._native.setExtension(this._handle, planes.length ? RayTracer.flattenPlanes(planes) : null, planes.length, ph, ordered.length);
            this._hasExtension = planes.length > 0 || phong;
        }
        const bg = this.backgroundColor;
        this._native.setBackground(this._handle, bg.x, bg.y, bg.z);
        if (useCamera && this.camera) {
            const b = this.camera.basis();
            this._native.setCamera(this._handle, Float64Array.of(
                b.origin.x, b.origin.y, b.origin.z, b.right.x, b.right.y, b.right.z,
                b.up.x, b.up.y, b.up.z, b.forward.x, b.forward.y, b.forward.z, b.tanHalfFov));
        } else {
            this._native.setCamera(this._handle, null);      // the reference's fixed camera
        }
    }

    static _dims(width, height, startY, scanHeight) {
        if (startY === undefined) startY = 0;              // RayTracer.js:134-140
        if (scanHeight === undefined) scanHeight = height;
        for (const [k, v] of [['width', width], ['height', height], ['startY', startY], ['scanHeight', scanHeight]]) {
            if (typeof v !== 'number' || !Number.isFinite(v) || Math.floor(v) !== v)
                throw new TypeError(`${k} must be a finite integer (got ${v})`);
        }
        if (width < 1 || height < 1 || scanHeight < 1 || width * scanHeight * 4 > 0x7fffffff)
            throw new RangeError('width, height, scanHeight must be >= 1 and width*scanHeight*4 < 2^31');
        return [width, height, startY, scanHeight];
    }

    /**
     * Same result bytes as the reference, including its row indexing quirk (RayTracer.js:169):
     * row y lands at row y-startY-1, the band's first row is dropped, its last stored row is zero.
     */
    render(width, height, startY, scanHeight, isFirstFrame) {
        const d = RayTracer._dims(width, height, startY, scanHeight);
        this._upload(false);
        return this._native.render(this._handle, d[0], d[1], d[2], d[3], this._native.QUIRK_ROW_SHIFT | this._modeFlags);
    }

    /**
     * render() into memory the caller already has: `target` is an ArrayBuffer or a typed array of at least
     * width * scanHeight * 4 bytes; returns it.  Same bytes as render() (row quirk included).  For loops that render frame
     * after frame: a fresh ArrayBuffer per frame (allocation, first-touch page faults, garbage collection) costs more than
     * a 1080p frame does.  options: {startY, scanHeight, exact: false -> every row at its true position, maxDepth, spp, seed}
     */
    renderInto(target, width, height, options = {}) {
        const d = RayTracer._dims(width, height, options.startY, options.scanHeight);
        const plain = options.exact === false || options.maxDepth || (options.spp || 1) > 1;     // the extension modes have no row quirk
        const e = RayTracer._ext(options);
        this._upload(!!options.useCamera);
        return this._native.renderInto(this._handle, target, d[0], d[1], d[2], d[3],
                                       (plain ? 0 : this._native.QUIRK_ROW_SHIFT) | this._modeFlags, e[0], e[1], e[2]);
    }

    /** Promise<ArrayBuffer> form of render() (runs off the main thread). */
    renderAsync(width, height, startY, scanHeight) {
        const d = RayTracer._dims(width, height, startY, scanHeight);
        this._upload(false);
        return this._native.renderAsync(this._handle, d[0], d[1], d[2], d[3], this._native.QUIRK_ROW_SHIFT | this._modeFlags);
    }

    static _ext(options) {
        const maxDepth = options.maxDepth === undefined ? 0 : options.maxDepth;
        const spp = options.spp === undefined ? 1 : options.spp;
        const seed = options.seed === undefined ? 0 : options.seed;
        if (!Number.isInteger(maxDepth) || maxDepth < 0 || maxDepth > 16) throw new RangeError('maxDepth must be an integer in [0, 16]');
        if (!Number.isInteger(spp) || spp < 1 || spp > 65536) throw new RangeError('spp must be an integer in [1, 65536]');
        if (!Number.isInteger(seed) || seed < 0 || seed > 0xffffffff) throw new RangeError('seed must be an unsigned 32-bit integer');
        return [maxDepth, spp, seed];
    }

    /** Every row at its true position.  options: {startY, scanHeight, useCamera, maxDepth, spp, seed} */
    renderFrame(width, height, options = {}) {
        const d = RayTracer._dims(width, height, options.startY, options.scanHeight);
        const e = RayTracer._ext(options);
        this._upload(!!options.useCamera);
        return this._native.render(this._handle, d[0], d[1], d[2], d[3], this._modeFlags, e[0], e[1], e[2]);
    }

    /** Promise<ArrayBuffer> form of renderFrame(). */
    renderFrameAsync(width, height, options = {}) {
        const d = RayTracer._dims(width, height, options.startY, options.scanHeight);
        const e = RayTracer._ext(options);
        this._upload(!!options.useCamera);
        return this._native.renderAsync(this._handle, d[0], d[1], d[2], d[3], this._modeFlags, e[0], e[1], e[2]);
    }

    /** Many rays at once: Float64Array origins (or null = (0,0,0)) and dirs, 3 numbers per ray. */
    traceBatch(origins, dirs) {
        this._upload(false);
        return this._native.traceBatch(this._handle, origins, dirs);
    }

    /**
     * The render cache's Sample.doRaytracing, batched (rendercache-js/src/Sample.js:49-75 -> Raytracer.js:24-123):
     * every ray starts at `from` (a Vector3, the cache's camera.from); returns {rgb, rgb8, hit, normal} with 3
     * numbers per ray — hit = the hit point or from + dir * 1e5 on a miss, normal = sample.normalDir (the outward
     * normal, or -hit on a miss).
     */
    traceSamples(from, dirs) {
        this._upload(false);
        return this._native.traceSamples(this._handle, Float64Array.of(from.x, from.y, from.z), dirs);
    }

    /** trace(rayOrigin, rayDir, depth) -> Vector3; depth is ignored, as in the reference (F1). */
    trace(rayOrigin, rayDir, depth) {
        const r = this.traceBatch(Float64Array.of(rayOrigin.x, rayOrigin.y, rayOrigin.z),
                                  Float64Array.of(rayDir.x, rayDir.y, rayDir.z));
        return new Vector3(r.rgb[0], r.rgb[1], r.rgb[2]);
    }

    /** requestPixel(rayDir) -> Color with Math.round(min(1,c)*255) per channel (RayTracer.js:98-115) */
    requestPixel(rayDir) {
        const c = this.trace(new Vector3(0, 0, 0), rayDir, 0);
        const color = new Color();
        color.r = Math.round(Math.min(1, c.x) * 255);
        color.g = Math.round(Math.min(1, c.y) * 255);
        color.b = Math.round(Math.min(1, c.z) * 255);
        return color;
    }

    /** work counters of the last call (retraced = pixels the hybrid mode re-traced in float64) */
    stats() { return this._native.stats(this._handle); }
}
