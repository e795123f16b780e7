// Soak of the render cache's frame loop: RenderCacheDriver with the cache resident on the GPU (ptr_rc_next_frame) against
// the same driver's host walks (statement for statement the reference's Driver.nextFrame, pinned to it by the goldens of
// run_tests.mjs) on random set-ups — frame sizes (odd ones, one-row ones), sample ratios, cache factors down to 0.5, the
// driver's tunables, random camera paths with jumps, restarts — comparing statistics, the four frames and every array of the
// state after every frame.
//   node test/soak_rc_loop.mjs [TRIALS] [SEED] [FIRST_TRIAL]          (PTR_SOAK_VERBOSE=1: one line per set-up and frame)
import assert from 'assert';
import { Vector3, Material, Sphere, Scene, RayTracer, RenderCacheDriver, RcCamera } from '../index.js';
import { perspectiveView } from '../examples/common.js';

const trials = Number(process.argv[2] || 100), seed0 = Number(process.argv[3] || 1), first = Number(process.argv[4] || 0);
const verbose = !!process.env.PTR_SOAK_VERBOSE, scale = Number(process.env.PTR_SOAK_SCALE || 1);      // PTR_SOAK_SCALE=8: frames up to 2 464 x 832

function mulberry(seed) {
    let a = seed >>> 0;
    return () => { a = (a + 0x6D2B79F5) >>> 0; let t = Math.imul(a ^ (a >>> 15), a | 1); t ^= t + Math.imul(t ^ (t >>> 7), t | 61); return ((t ^ (t >>> 14)) >>> 0) / 4294967296; };
}

function randomScene(r) {
    const u = (a, b) => a + (b - a) * r(), v = (x, y, z) => new Vector3(x, y, z);
    const s = new Scene();
    s.add(new Sphere(v(0, -10004, -20), 10000, new Material(v(u(.1, .9), u(.1, .9), u(.1, .9)), 0, 0, v(0, 0, 0))));
    const n = 2 + Math.floor(r() * 9);
    for (let i = 0; i < n; i++)
        s.add(new Sphere(v(u(-10, 10), u(-3, 6), u(-40, -10)), u(0.5, 4), new Material(v(u(0, 1), u(0, 1), u(0, 1)), r() < .5 ? 1 : 0, r() < .3 ? .5 : 0, v(0, 0, 0))));
    for (let i = 0; i < 1 + Math.floor(r() * 2); i++)
        s.add(new Sphere(v(u(-20, 20), u(8, 30), u(-40, 15)), u(0.5, 3), new Material(v(0, 0, 0), 0, 0, v(u(.3, 1.5), u(.3, 1.5), u(.3, 1.5)))));
    return s;
}

const arrays = ['itemPixel', 'itemHit', 'itemAge', 'itemRgb', 'pixelElement', 'pixelDepth', 'pixelRgb', 'pixelPriority', 'pixelSample'];
const same = (a, b) => { for (let i = 0; i < a.length; i++) if (a[i] !== b[i]) return false; return true; };
const realRandom = Math.random;
let frames = 0, pixels = 0, reusedAll = 0, refused = 0;
const t0 = Date.now();
for (let t = first; t < trials; t++) {
    const r = mulberry(seed0 * 100003 + t);
    const pick = a => a[Math.floor(r() * a.length)];
    const w = scale * (r() < 0.1 ? 8 + Math.floor(r() * 300) : 8 + Math.floor(r() * 140)) + (scale > 1 ? Math.floor(r() * scale) : 0);
    const h = r() < 0.05 ? 1 + Math.floor(r() * 3) : scale * (4 + Math.floor(r() * 100)) + (scale > 1 ? Math.floor(r() * scale) : 0);
    const ratio = pick([1, 2, 3, 4, 8, 16, 32]), nframes = 3 + Math.floor(r() * 6);
    const before = r() < 0.35 ? { cacheFactor: pick([0.5, 1, 1.5, 2]) } : {};
    if (before.cacheFactor && !Number.isInteger(w * h * before.cacheFactor)) before.cacheFactor = Math.ceil(before.cacheFactor);
    if (w * h * (before.cacheFactor || 4) < 8) before.cacheFactor = 4;                          // (a cache below one page: nothing to allocate from)
    const tweak = r() < 0.5 ? {} : { ageFactor: pick([1, 2, 0.5, 3]), interpolationZero: pick([20, 5, 60]), interpolationRandom: pick([5, 0, 1, 17]),
                                     priorityMax: pick([255, 100, 7, 40]), depthThreshold: pick([1e6, 60, 35]) };
    const bg = new Vector3(r() * 2, r() * 2, r() * 2);
    const drvs = [false, true].map(resident => {
        const rs = mulberry(seed0 * 7919 + t);                                               // the same scene twice
        const rt = new RayTracer(bg, randomScene(rs), undefined, { mode: 'exact' });
        const cam = new RcCamera(w, h), drv = new RenderCacheDriver(rt, cam, ratio, { resident });
        Object.assign(drv, before); drv.prepare(false); Object.assign(drv, tweak);
        return { rt, cam, drv };
    });
    let reused = 0;
    const clearItem = drvs[0].drv._clearItem;
    drvs[0].drv._clearItem = function (i) { reused++; return clearItem.call(this, i); };
    const a = new Uint32Array(w * h), b = new Uint32Array(w * h);
    const restartAt = r() < 0.15 ? 2 : -1;
    let from = [r() * 8 - 4, r() * 3, 2 + r() * 6], to = [r() * 6 - 3, 0, -20];
    const tag0 = `trial ${t} (seed ${seed0}): ${w}x${h} ratio ${ratio} ${JSON.stringify(before)} ${JSON.stringify(tweak)}`;
    if (verbose) console.log(tag0);
    for (let k = 0; k < nframes; k++) {
        const jump = r() < 0.15 ? 6 : 0.6;
        from = from.map(c => c + (r() - 0.5) * jump); to = to.map((c, i) => c + (i < 2 ? (r() - 0.5) * jump : 0));
        const view = perspectiveView(from, to, 30, w / h);
        const frameIndex = k === restartAt ? 0 : k, tag = `${tag0} frame ${k}`;
        const thrown = [];
        for (const d of drvs) {
            if (verbose) console.log(`  frame ${k} (index ${frameIndex}) ${d.drv.resident ? 'resident' : 'host'}`);
            Math.random = mulberry(seed0 * 31 + t * 17 + k);
            d.cam.setView(view.from, view.matrixWorld, view.matrixWorldInverse, view.projectionMatrix, view.projectionMatrixInverse);
            try { d.drv.nextFrame(frameIndex); thrown.push(null); } catch (e) { thrown.push(String(e.message)); }
        }
        Math.random = realRandom;
        if (thrown[0] !== null || thrown[1] !== null) {        // a frame the reference's pick loop would never leave: refused on both sides
            assert.ok(thrown[0] !== null && thrown[1] !== null, `${tag}: host ${thrown[0]} / resident ${thrown[1]}`);
            assert.ok(/would not end/.test(thrown[0]) && /would\s+not end/.test(thrown[1]), `${tag}: ${thrown[0]} / ${thrown[1]}`);
            refused++;
            break;
        }
        const [host, gpu] = drvs.map(d => d.drv);
        assert.deepStrictEqual(gpu.statistics, host.statistics, tag + ' statistics');
        assert.strictEqual(gpu.getCacheUsage(), host.getCacheUsage(), tag + ' cache usage');
        assert.strictEqual(gpu.totalPriority, host.totalPriority, tag + ' totalPriority');
        for (const f of ['getColorFrame', 'getReprojectionFrame', 'getPriorityFrame', 'getSamplingFrame'])
            assert.ok(same(gpu[f](a), host[f](b)), `${tag} ${f}`);
        gpu.syncToHost();
        const pages = Math.floor(host.cacheSize / 8);
        assert.strictEqual((gpu.cachePointer / 8) % pages, (host.cachePointer / 8) % pages, tag + ' cachePointer');
        for (const name of arrays) {
            const x = gpu[name], y = host[name];
            let bad = -1;
            for (let i = 0; i < x.length && bad < 0; i++) if (!(x[i] === y[i] || (x[i] !== x[i] && y[i] !== y[i]))) bad = i;
            assert.strictEqual(bad, -1, `${tag} ${name}[${bad}]: ${x[bad]} !== ${y[bad]}`);
        }
        frames++; pixels += w * h;
    }
    reusedAll += reused;
    for (const d of drvs) { d.drv.dispose(); d.rt.dispose(); }
    if ((t + 1) % (scale > 1 ? 2 : 20) === 0) console.log(`  ${t + 1} trials, ${frames} frames, ${((Date.now() - t0) / 1000).toFixed(0)} s`);
}
console.log(`soak_rc_loop seed ${seed0}: ${trials} set-ups, ${frames} frames = ${(pixels / 1e6).toFixed(1)} M pixels, ${reusedAll} items reused from full pages, ` +
            `${refused} set-ups ended by a frame that both forms refuse (no sample counted: the reference's pick loop would not end); ` +
            'statistics, frames and every array of the state equal after every frame');
