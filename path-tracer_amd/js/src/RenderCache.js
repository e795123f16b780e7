import native from './native.js';

/**
 * RenderCache — GPU passes of the render cache (rendercache-js/src/Driver.js).  Built so far:
 * reprojectFrame, the drop-in for Driver.reprojectFrame() (Driver.js:417-486, with detachFrameFromCache :405-416
 * and Camera.reprojectPixel, Camera.js:52-74): same float64 arithmetic, same results as the reference's sequential
 * walk over the cache, computed in parallel on an MI355X.
 *
 *   const rc = new RenderCache({device: 0});
 *   driver.reprojectFrame = () => rc.reprojectDriver(driver);      // reference-shaped Driver object
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
}
