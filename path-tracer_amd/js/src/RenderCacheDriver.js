import RenderCache from './RenderCache.js';

const INTERPOLATED = 0, URGENT = 1, SAMPLED = 2;
const NEW_PIXEL_DEPTH = 10000000000000.0;          // Pixel.js:8

/**
 * RcCamera — what the render cache needs from its camera (rendercache-js/src/Camera.js), without three.js: the caller
 * hands over the matrices its own camera holds (a THREE.PerspectiveCamera's matrixWorld, matrixWorldInverse,
 * projectionMatrix, projectionMatrixInverse `.elements`, column-major) and the eye position.
 */
export class RcCamera {
    constructor(width, height) {
        this.scope = { x: width, y: height };
        this.from = { x: 0, y: 0, z: 0 };
        this.matrixWorld = new Float64Array(16); this.matrixWorldInverse = new Float64Array(16);
        this.projectionMatrix = new Float64Array(16); this.projectionMatrixInverse = new Float64Array(16);
    }

    setView(from, matrixWorld, matrixWorldInverse, projectionMatrix, projectionMatrixInverse) {
        this.from = { x: from[0], y: from[1], z: from[2] };
        this.matrixWorld.set(matrixWorld); this.matrixWorldInverse.set(matrixWorldInverse);
        this.projectionMatrix.set(projectionMatrix); this.projectionMatrixInverse.set(projectionMatrixInverse);
    }

    /**
     * Camera.computeDirToXY (Camera.js:39-50): THREE.Raycaster.setFromCamera for a perspective camera — the point
     * (ndc.x, ndc.y, 0.5) unprojected (projectionMatrixInverse, then matrixWorld, each with its divide), minus the
     * camera's world position, normalized by three.js and once more by the cache's own Vector3.  Writes 3 numbers.
     */
    dirToXY(x, y, out, at) {
        let vx = (x / this.scope.x) * 2 - 1, vy = (y / this.scope.y) * 2 - 1, vz = 0.5;
        for (const e of [this.projectionMatrixInverse, this.matrixWorld]) {      // Vector3.applyMatrix4, twice
            const w = 1 / (e[3] * vx + e[7] * vy + e[11] * vz + e[15]);
            const nx = (e[0] * vx + e[4] * vy + e[8] * vz + e[12]) * w;
            const ny = (e[1] * vx + e[5] * vy + e[9] * vz + e[13]) * w;
            const nz = (e[2] * vx + e[6] * vy + e[10] * vz + e[14]) * w;
            vx = nx; vy = ny; vz = nz;
        }
        const m = this.matrixWorld;
        vx -= m[12]; vy -= m[13]; vz -= m[14];
        const s = 1 / (Math.sqrt(vx * vx + vy * vy + vz * vz) || 1);             // THREE.Vector3.normalize
        vx *= s; vy *= s; vz *= s;
        const len2 = vx * vx + vy * vy + vz * vz;                                // Vector3.normalize of the cache (Vector3.js:23-33)
        if (len2 > 0) { const inv = 1 / Math.sqrt(len2); vx *= inv; vy *= inv; vz *= inv; }
        out[at] = vx; out[at + 1] = vy; out[at + 2] = vz;
    }
}

/**
 * RenderCacheDriver — the render cache's frame loop (rendercache-js/src/Driver.js) on an MI355X.  The per-sample trace,
 * the reprojection and the depth culling / gap filling run on the GPU (ptr_trace_samples_f64, ptr_rc_reproject,
 * ptr_rc_shade); what the reference does sequentially between them — the priority threshold, the serpentine candidate
 * walk with its error diffusion, the paged cache allocator, ageing (Driver.js:252-326, :682-839, :936-942) — stays on
 * the host, restated here over flat typed arrays instead of Pixel / Sample objects.  Same method names as the reference
 * where a caller would use them: prepare, nextFrame, getColorFrame, getReprojectionFrame, getPriorityFrame,
 * getSamplingFrame, getCacheUsage, statistics.  Pixels live on the (w + 2) x (h + 2) grid of Driver.allocBuffers.
 *
 *   const drv = new RenderCacheDriver(rayTracer, new RcCamera(w, h), 16);   // rayTracer: ../RayTracer.js with a scene
 *   const drv = new RenderCacheDriver(rayTracer, camera, 16, { resident: true });
 *       // the cache and the pixel grid stay in HBM (ptr_rc_cache_*, ptr_rc_next_frame): per frame the camera goes down and the
 *       // statistics come back, the host walks of this file run as GPU passes with the same results; only the first frame's
 *       // initializeCache (Math.random) runs here.  syncToHost() brings the state back into the typed arrays below.
 *   drv.prepare(false);
 *   camera.setView(from, cam.matrixWorld.elements, ...); drv.nextFrame(k); drv.getColorFrame(rgba);
 */
export default class RenderCacheDriver {
    constructor(engine, camera, ratio, options = {}) {
        this.engine = engine; this.camera = camera;
        this.rc = new RenderCache({ device: options.device | 0 });
        this.resident = !!options.resident;
        this._onGpu = false;                                     // the state lives in the engine's GPU context
        this.test = false;
        this.cachePointer = 0; this.cachePageSize = 8;
        this.initialFill = 0.2;
        this.maximumSamplesPerFrameRatio = 1 / ratio;
        this.maximumSamplesPerFrame = camera.scope.x * camera.scope.y * this.maximumSamplesPerFrameRatio;
        this.interpolationRandom = 5; this.interpolationZero = 20;
        this.cacheFactor = 4; this.depthThreshold = 1e6; this.ageFactor = 1;
        this.priorities = [new Array(256).fill(0), new Array(256).fill(0), new Array(256).fill(0)];
        this.priorityLevels = 256; this.totalPriority = 0; this.priorityMax = 255;
        this.statistics = { frameIndex: 0, startingCacheUsage: 0, iterationsToFill: 0, initialFill: 0, completeness: 0,
                            totalPriority: 0, requestFactor: 0, threshold: 0, candidates: 0, requests: 0 };
    }

    dispose() {
        if (this._onGpu && this.engine._handle) this.engine._native.rcCacheDestroy(this.engine._handle);
        this._onGpu = false;
        this.rc.dispose();
    }

    // ---- resident mode: the state's way to and from the GPU ----------------------------------------------------------
    _stateArrays(priority) {
        return [this.itemPixel, this.itemHit, this.itemAge, this.itemRgb, this.pixelElement, this.pixelDepth, this.pixelRgb,
                priority, this.pixelSample];
    }

    _toGpu() {
        const nat = this.engine._native, h = this.engine._handle;
        if (!h) throw new Error('RayTracer was disposed');
        nat.rcCacheCreate(h, this.camera.scope.x, this.camera.scope.y, this.cacheSize);
        nat.rcCacheUpload(h, ...this._stateArrays(Int32Array.from(this.pixelPriority)), this.cachePointer);
        this._onGpu = true;
    }

    /** resident mode: copy the cache and the pixel grid back into this object's typed arrays (the GPU copy stays in charge) */
    syncToHost() {
        if (!this._onGpu) return;
        const prio = new Int32Array(this.pixelPriority.length);
        this.cachePointer = this.engine._native.rcCacheDownload(this.engine._handle, ...this._stateArrays(prio));
        this.pixelPriority.set(prio);
    }

    _residentFrame() {
        const nat = this.engine._native, cam = this.camera;
        if (!this._onGpu) this._toGpu();
        this.engine._upload(false);                              // the scene is re-read every frame, as traceSamples does
        const view = new Float64Array(67);
        view.set([cam.from.x, cam.from.y, cam.from.z], 0);
        view.set(cam.matrixWorld, 3); view.set(cam.matrixWorldInverse, 19);
        view.set(cam.projectionMatrix, 35); view.set(cam.projectionMatrixInverse, 51);
        const st = nat.rcNextFrame(this.engine._handle, view, Float64Array.of(this.ageFactor, this.depthThreshold, this.interpolationZero,
                                   this.interpolationRandom, this.priorityMax, this.maximumSamplesPerFrame));
        const s = this.statistics;
        s.threshold = Number.isNaN(st[0]) ? null : st[0];
        s.requestFactor = st[1]; s.candidates = st[2]; s.requests = st[3]; s.completeness = st[4]; s.totalPriority = st[5];
        this.totalPriority += st[6];
        this._cacheUsage = st[7];
    }

    /** Driver.prepare (:81-88): buffers first, cache next, and only then the test flag (so cacheFactor stays 4) */
    prepare(isTest) {
        const w = this.camera.scope.x, h = this.camera.scope.y;
        this.stride = w + 2;
        const cells = this.stride * (h + 2);
        this.pixelElement = new Int32Array(cells).fill(-1);     // pixel.element (cache index)
        this.pixelDepth = new Float64Array(cells).fill(NEW_PIXEL_DEPTH);
        this.pixelRgb = new Uint32Array(cells);                 // pixel.color, r | g << 8 | b << 16
        this.pixelPriority = new Float64Array(cells);
        this.pixelSample = new Uint8Array(cells);
        this.cacheSize = w * h * this.cacheFactor;
        const n = this.cacheSize;
        this.itemPixel = new Int32Array(n).fill(-1);            // sample.pixel (grid index)
        this.itemHit = new Float64Array(3 * n).fill(NaN);       // sample.hit, NaN = null
        this.itemNormal = new Float64Array(3 * n);
        this.itemAge = new Float64Array(n);                     // Sample starts with age 0 (Sample.js:10)
        this.itemRgb = new Uint32Array(n);
        this.test = isTest;
        if (isTest) this.cacheFactor = 1.1;
    }

    // ---- the paged allocator (Driver.allocCacheItem :252-305) ------------------------------------------------------
    _clearItem(i) {                                             // Sample.clear(): the old pixel keeps pointing at the item
        this.itemPixel[i] = -1; this.itemHit[3 * i] = NaN; this.itemAge[i] = -1; this.itemRgb[i] = 0;
    }

    allocCacheItem(pixel) {
        let stop = this.cachePointer + this.cachePageSize;
        if (stop > this.cacheSize) stop = this.cachePageSize;
        let maxAge = -1, maxAgeIndex = -1;
        for (let i = stop - this.cachePageSize; i < stop; i++) {
            if (this.itemPixel[i] < 0) {                        // empty: take it
                this.itemPixel[i] = pixel; this.pixelElement[pixel] = i;
                this.cachePointer = stop;
                return i;
            }
            if (this.itemAge[i] > maxAge) { maxAge = this.itemAge[i]; maxAgeIndex = i; }
        }
        // the page is full: its oldest item is reused.  (All ages -1 cannot happen for attached items: clear() detaches.)
        this._clearItem(maxAgeIndex);
        this.itemPixel[maxAgeIndex] = pixel; this.pixelElement[pixel] = maxAgeIndex;
        this.cachePointer = stop;
        return maxAgeIndex;
    }

    addRequest(requests, pixel) {                               // :311-326
        this.pixelSample[pixel] = 1;
        let item = this.pixelElement[pixel];
        if (item < 0) item = this.allocCacheItem(pixel);
        requests.push(item);
    }

    /** Driver.initializeCache (:328-353): random pixels, x and y drawn from 0 (so border cells get samples too) */
    initializeCache() {
        const w = this.camera.scope.x, h = this.camera.scope.y;
        let iterations = 0, cacheUsage = 0.0;
        this.statistics.initialFill = this.initialFill;
        while (cacheUsage < this.initialFill) {
            const requests = [];
            for (let i = 0; i < this.maximumSamplesPerFrame; i++) {
                const x = Math.round(Math.random() * (w - 1)), y = Math.round(Math.random() * (h - 1));
                this.addRequest(requests, y * this.stride + x);
            }
            this.requestSamples(requests);
            this.age(this.ageFactor, requests);
            cacheUsage += this.maximumSamplesPerFrameRatio;
            iterations++;
        }
        this.statistics.startingCacheUsage = cacheUsage * 100.0;
        this.statistics.iterationsToFill = iterations;
    }

    initializeCacheWithEntireFrame() {                          // :356-371
        const requests = [];
        for (let y = 1; y <= this.camera.scope.y; y++)
            for (let x = 1; x <= this.camera.scope.x; x++) this.addRequest(requests, y * this.stride + x);
        this.requestSamples(requests);
        this.age(this.ageFactor, requests);
    }

    /** Driver.nextFrame (:95-115) */
    nextFrame(frameIndex) {
        if (this.test) { this.initializeCacheWithEntireFrame(); return; }
        if (frameIndex === 0) {
            if (this._onGpu) { this.syncToHost(); this._onGpu = false; }      // a restart: the cache is filled on the host again
            this.initializeCache();
        }
        this.frameIndex = frameIndex;
        if (this.resident) return this._residentFrame();
        this.reprojectFrame();
        this.shadeFrame();                                      // depthCulling + fillGaps
        const requests = this.directSamples();
        this.requestSamples(requests);
        for (let i = 0; i < this.cacheSize; i++) this.itemAge[i] += this.ageFactor;      // age(ageFactor, cache)
    }

    /**
     * Driver.nextFrame1SPP (:117-128; what the reference's index.js calls when its `raytrace` switch is on, with ratio 1): gap
     * filling, selection and trace alone - no reprojection, no depth culling, no ageing, the grid keeps its elements.  Host form
     * (a resident cache is brought back first and goes up again with the next nextFrame()).
     */
    nextFrame1SPP(frameIndex) {
        if (this.test) { this.initializeCacheWithEntireFrame(); return; }
        if (this._onGpu) { this.syncToHost(); this._onGpu = false; }
        if (frameIndex === 0) this.initializeCache();
        this.shadeFrame(true);                                  // fillGaps
        const requests = this.directSamples();
        this.requestSamples(requests);
    }

    /** Driver.reprojectFrame (:417-486) on the GPU; the border of the grid is not touched, as in the reference */
    reprojectFrame() {
        const cam = this.camera, w = cam.scope.x, h = cam.scope.y, st = this.stride;
        const out = this.rc.reprojectFrame({ from: [cam.from.x, cam.from.y, cam.from.z], view: cam.matrixWorldInverse,
                                             proj: cam.projectionMatrix, width: w, height: h },
                                           this.itemHit, this.itemAge, this.ageFactor, this.depthThreshold);
        this.itemPixel.set(out.itemPixel);
        for (let y = 1; y <= h; y++) {
            const a = y * st + 1, b = a + w;
            this.pixelElement.set(out.pixelElement.subarray(a, b), a);
            this.pixelDepth.set(out.pixelDepth.subarray(a, b), a);
            this.pixelSample.fill(0, a, b);                     // detachFrameFromCache (:408)
        }
    }

    /** Driver.depthCulling + Driver.fillGaps (:488-679) on the GPU */
    shadeFrame(fillOnly = false) {
        const w = this.camera.scope.x, h = this.camera.scope.y, st = this.stride;
        // fillGaps alone: the culling step of the pass adds 10 age factors where depths disagree - with a factor of 0 it adds
        // nothing, and the colour it copies to a pixel from its element is the colour that pixel already has (requestSamples
        // stores both, Sample.js:68)
        const out = this.rc.shadeFrame(w, h, this.pixelElement, this.pixelDepth, this.itemAge, this.itemRgb,
                                       { ageFactor: fillOnly ? 0 : this.ageFactor, interpolationZero: this.interpolationZero,
                                         interpolationRandom: this.interpolationRandom, priorityMax: this.priorityMax }, this.pixelRgb);
        for (let y = 1; y <= h; y++) {
            const a = y * st + 1, b = a + w;
            this.pixelRgb.set(out.pixelRgb.subarray(a, b), a);
            for (let p = a; p < b; p++) this.pixelPriority[p] = out.pixelPriority[p];
        }
        for (let k = 0; k < 3; k++) for (let j = 0; j < 256; j++) this.priorities[k][j] = out.hist[256 * k + j];
        this.statistics.completeness = out.stats[0];
        this.statistics.totalPriority = out.stats[1];
        this.totalPriority += out.stats[2];
        this.reprojectionFrame = out.reprojectionFrame;         // as of fillGaps (the reference's getReprojectionFrame reads live state)
    }

    /** Driver.directSamples (:682-827), statement for statement where the order of effects matters */
    directSamples() {
        const w = this.camera.scope.x, h = this.camera.scope.y, st = this.stride, prio = this.pixelPriority;
        let samples = null, threshold = null;
        for (let i = this.priorityMax; i >= 0; i--) {
            let test = this.priorities[URGENT][i] + this.priorities[INTERPOLATED][i] + this.priorities[SAMPLED][i];
            if (samples === null) { samples = test; test = 0; }
            if (samples + test > this.maximumSamplesPerFrame) { threshold = i; samples += test; break; }
            samples += test;
        }
        const requestFactor = samples / this.maximumSamplesPerFrame;
        this.statistics.threshold = threshold;
        this.statistics.requestFactor = requestFactor;

        // candidates, rows alternately left-to-right and right-to-left (`>= null` compares with 0, as in the reference)
        const candPixel = [], candDx = [];
        let last = -1;                                          // the pixel the reference's `var pixel` is left holding
        for (let y = 1; y <= h; y++) {
            if (y % 2 === 0) {
                for (let x = 1; x <= w; x++) { last = y * st + x; if (prio[last] >= threshold) { candPixel.push(last); candDx.push(1); } }
            } else {
                for (let x = w; x > 0; x--) { last = y * st + x; if (prio[last] >= threshold) { candPixel.push(last); candDx.push(-1); } }
            }
        }
        this.statistics.candidates = candPixel.length;
        // (no sample counted at any level up to priorityMax: the reference's loop below adds 0 to `index` for ever; refused here
        // as the resident form refuses it, ptr_rc_next_frame)
        if (candPixel.length > 0 && (!(requestFactor > 0) || candPixel.length / requestFactor > 4.0e9))
            throw new RangeError(`request factor ${requestFactor} with ${candPixel.length} candidates: the reference's pick loop ` +
                                 '(Driver.js:795-822) would not end');

        const requests = [];
        let index = 0, lastI = null;
        while (index < candPixel.length) {
            const i = Math.round(index);
            if (lastI === null || lastI !== i) {
                if (i < candPixel.length) {
                    const p = candPixel[i];
                    this.addRequest(requests, p);
                    // distributeExceedingPriority (:829-839) is handed the STALE `pixel` of the candidate loops (:813-814):
                    // the excess of that one pixel is spread to the right/left and lower neighbours of every picked candidate
                    const half = Math.floor((prio[last] - threshold) / 2.0);
                    if (half >= 0) { prio[p + candDx[i]] += half; prio[p + st] += half; }
                }
            }
            lastI = i;
            index += requestFactor;
        }
        return requests;
    }

    /** Driver.requestSamples (:842-863) with Sample.doRaytracing (Sample.js:49-75), all rays of the frame in one batch */
    requestSamples(requests) {
        this.statistics.requests = requests.length;
        const live = requests.filter(it => this.itemPixel[it] >= 0), n = live.length;
        if (n === 0) return;
        const dirs = new Float64Array(3 * n), st = this.stride;
        for (let k = 0; k < n; k++) {                           // Sample.resample is never set: always the pixel's direction
            const p = this.itemPixel[live[k]];
            this.camera.dirToXY(p % st, Math.floor(p / st), dirs, 3 * k);
        }
        const r = this.engine.traceSamples(this.camera.from, dirs);
        for (let k = 0; k < n; k++) {                           // in request order: a pixel sampled twice keeps the last colour
            const it = live[k];
            const cr = Math.round(Math.min(1, r.rgb[3 * k]) * 255), cg = Math.round(Math.min(1, r.rgb[3 * k + 1]) * 255),
                  cb = Math.round(Math.min(1, r.rgb[3 * k + 2]) * 255);
            this.itemRgb[it] = (cr & 255) | ((cg & 255) << 8) | ((cb & 255) << 16);
            for (let c = 0; c < 3; c++) { this.itemHit[3 * it + c] = r.hit[3 * k + c]; this.itemNormal[3 * it + c] = r.normal[3 * k + c]; }
            this.pixelRgb[this.itemPixel[it]] = this.itemRgb[it];
        }
    }

    age(amount, items) { for (let i = 0; i < items.length; i++) this.itemAge[items[i]] += amount; }     // :936-942

    getCacheUsage() {
        if (this._onGpu) return this._cacheUsage;
        let used = 0;
        for (let i = 0; i < this.cacheSize; i++) if (this.itemPixel[i] >= 0) used++;
        return used / this.cacheSize;
    }

    // ---- frames (:960-1046): rows bottom-up, RGBA8 --------------------------------------------------------------------
    _gpuFrame(which, buf) {
        const out = buf instanceof Uint32Array ? buf : new Uint32Array(this.camera.scope.x * this.camera.scope.y);
        this.engine._native.rcGetFrame(this.engine._handle, which, out);
        if (out !== buf) for (let k = 0; k < out.length; k++) buf[k] = out[k];
        return buf;
    }
    _frame(buf, f) {
        const w = this.camera.scope.x, h = this.camera.scope.y, st = this.stride;
        let k = 0;
        for (let y = h - 1; y >= 0; y--) for (let x = 0; x < w; x++, k++) buf[k] = f((y + 1) * st + x + 1);
        return buf;
    }
    getColorFrame(buf) { return this._onGpu ? this._gpuFrame(0, buf) : this._frame(buf, p => ((255 << 24) | this.pixelRgb[p]) >>> 0); }
    getReprojectionFrame(buf) {
        return this._onGpu ? this._gpuFrame(1, buf) : this._frame(buf, p => ((255 << 24) | (this.pixelElement[p] >= 0 ? this.itemRgb[this.pixelElement[p]] : 0)) >>> 0);
    }
    getPriorityFrame(buf) {
        return this._onGpu ? this._gpuFrame(2, buf) : this._frame(buf, p => { const v = this.pixelPriority[p]; return ((255 << 24) | (v << 16) | (v << 8) | v) >>> 0; });
    }
    getSamplingFrame(buf) { return this._onGpu ? this._gpuFrame(3, buf) : this._frame(buf, p => (this.pixelSample[p] ? 0xffffffff : 0xff000000) >>> 0); }
}
