import native from './native.js';

/**
 * RenderCache — GPU passes of the render cache (rendercache-js/src/Driver.js).  Built so far:
 * reprojectFrame, the drop-in for Driver.reprojectFrame() (Driver.js:417-486, with detachFrameFromCache :405-416
 * and Camera.reprojectPixel, Camera.js:52-74), and shadeFrame, the drop-in for Driver.depthCulling() + Driver.fillGaps()
 * (:488-679) with the frame getters (:960-1018): same arithmetic, same results as the reference's sequential walks,
 * computed in parallel on an MI355X.
 *
 *   const rc = new RenderCache({device: 0});
 *   driver.reprojectFrame = () => rc.reprojectDriver(driver);      // reference-shaped Driver object
 *   driver.depthCulling = () => {}; driver.fillGaps = () => rc.shadeDriver(driver);
 */
export default class RenderCache {
    constructor(options = {}) {
        this._native = native();
        this._handle = this._native.create(options.device | 0);
    }

    dispose() {
        if (this._handle) { this._native.destroy(this._handle); this._handle = null; }
    }

    /**
     * camera: {from: [x,y,z], view: 16 numbers (matrixWorldInverse.elements), proj: 16 numbers
     * (projectionMatrix.elements), width, height};  hits: Float64Array(3n), NaN in x = the item has no hit;
     * age: Float64Array(n), updated in place.  Returns {itemPixel, pixelElement, pixelDepth}: pixels are addressed
     * y * (width + 2) + x on the padded grid of Driver.allocBuffers.
     */
    reprojectFrame(camera, hits, age, ageFactor = 1, depthThreshold = 1e6) {
        if (!this._handle) throw new Error('RenderCache was disposed');
        const cam = new Float64Array(35);
        cam.set(camera.from, 0); cam.set(camera.view, 3); cam.set(camera.proj, 19);
        return this._native.rcReproject(this._handle, cam, camera.width, camera.height, hits, age, ageFactor, depthThreshold);
    }

    /**
     * Drop-in body for Driver.reprojectFrame(): reads driver.cache[i].hit / .age, driver.camera (from, scope,
     * cam.matrixWorldInverse, cam.projectionMatrix), driver.ageFactor / depthThreshold, and writes back
     * cacheItem.age, cacheItem.pixel, pixel.element, pixel.depth, pixel.sample exactly as the reference leaves them.
     */
    reprojectDriver(driver) {
        const n = driver.cacheSize, cam = driver.camera;
        const hits = new Float64Array(3 * n), age = new Float64Array(n);
        for (let i = 0; i < n; i++) {
            const it = driver.cache[i];
            if (it.hit !== null) { hits[3 * i] = it.hit.x; hits[3 * i + 1] = it.hit.y; hits[3 * i + 2] = it.hit.z; }
            else hits[3 * i] = NaN;
            age[i] = it.age;
        }
        const w = cam.scope.x, h = cam.scope.y, stride = w + 2;
        const out = this.reprojectFrame({ from: [cam.from.x, cam.from.y, cam.from.z], view: cam.cam.matrixWorldInverse.elements,
                                          proj: cam.cam.projectionMatrix.elements, width: w, height: h },
                                        hits, age, driver.ageFactor, driver.depthThreshold);
        for (let i = 0; i < n; i++) {
            const it = driver.cache[i], p = out.itemPixel[i];
            it.age = age[i];
            it.pixel = p < 0 ? null : driver.getPixel(p % stride, Math.floor(p / stride));
        }
        for (let y = 1; y <= h; y++) for (let x = 1; x <= w; x++) {
            const pixel = driver.getPixel(x, y), e = out.pixelElement[y * stride + x];
            pixel.sample = false;                                          // detachFrameFromCache (Driver.js:409)
            pixel.depth = out.pixelDepth[y * stride + x];
            pixel.element = e < 0 ? null : driver.cache[e];
        }
    }

    /**
     * Depth culling + gap filling + frames on flat arrays (see include/ptr.h ptr_rc_shade).  itemAge is updated in place.
     * params: {ageFactor, interpolationZero, interpolationRandom, priorityMax};  pixelRgbIn (optional, the grid's colours
     * before the pass): only read for BORDER cells that carry an element (the reference never detaches the border).
     */
    shadeFrame(width, height, pixelElement, pixelDepth, itemAge, itemRgb, params = {}, pixelRgbIn = undefined) {
        if (!this._handle) throw new Error('RenderCache was disposed');
        const p = Float64Array.of(params.ageFactor === undefined ? 1 : params.ageFactor,
                                  params.interpolationZero === undefined ? 20 : params.interpolationZero,
                                  params.interpolationRandom === undefined ? 5 : params.interpolationRandom,
                                  params.priorityMax === undefined ? 255 : params.priorityMax);
        return this._native.rcShade(this._handle, width, height, pixelElement, pixelDepth, itemAge, itemRgb, p, pixelRgbIn);
    }

    /**
     * Drop-in body for Driver.depthCulling() followed by Driver.fillGaps(): reads pixel.element / pixel.depth and
     * cacheItem.age / .color of a reference-shaped Driver, writes back cacheItem.age, pixel.color (r, g, b), pixel.priority,
     * driver.priorities[0..2][0..255], driver.statistics.completeness / .totalPriority and driver.totalPriority.
     * Returns the three frames {colorFrame, reprojectionFrame, priorityFrame} (Uint32Array, rows bottom-up).
     * Pixels that share their element's Color object in the reference get their own {r, g, b} copy here.
     */
    shadeDriver(driver) {
        const cam = driver.camera, w = cam.scope.x, h = cam.scope.y, stride = w + 2, cells = stride * (h + 2), n = driver.cacheSize;
        const pel = new Int32Array(cells).fill(-1), pdep = new Float64Array(cells), age = new Float64Array(n), rgb = new Uint32Array(n);
        const index = new Map();
        for (let i = 0; i < n; i++) {
            const it = driver.cache[i];
            index.set(it, i);
            age[i] = it.age;
            if (it.color) rgb[i] = (it.color.r & 255) | ((it.color.g & 255) << 8) | ((it.color.b & 255) << 16);
        }
        for (let y = 1; y <= h; y++) for (let x = 1; x <= w; x++) {
            const px = driver.getPixel(x, y);
            pdep[y * stride + x] = px.depth;
            if (px.element) pel[y * stride + x] = index.get(px.element);
        }
        const out = this.shadeFrame(w, h, pel, pdep, age, rgb, { ageFactor: driver.ageFactor, interpolationZero: driver.interpolationZero,
                                    interpolationRandom: driver.interpolationRandom, priorityMax: driver.priorityMax });
        for (let i = 0; i < n; i++) driver.cache[i].age = age[i];
        for (let y = 1; y <= h; y++) for (let x = 1; x <= w; x++) {
            const px = driver.getPixel(x, y), c = out.pixelRgb[y * stride + x];
            px.color = px.element ? px.element.color : { r: c & 255, g: (c >> 8) & 255, b: (c >> 16) & 255 };
            px.priority = out.pixelPriority[y * stride + x];
        }
        for (let k = 0; k < 3; k++) { driver.priorities[k] = Array.from(out.hist.subarray(256 * k, 256 * k + 256)); }
        driver.statistics.completeness = out.stats[0];
        driver.statistics.totalPriority = out.stats[1];
        driver.totalPriority += out.stats[2];
        return { colorFrame: out.colorFrame, reprojectionFrame: out.reprojectionFrame, priorityFrame: out.priorityFrame };
    }
}
