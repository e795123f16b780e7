// TEST INFRASTRUCTURE — golden vectors of whole render-cache frames (SURVEY.md §8f rank 4: sample selection, the
// paged cache allocator, and the frame loop that ties ranks 1-3 together).
//
// Runs the UNMODIFIED reference under Node: rendercache-js/src/Driver.js (nextFrame :95-115 with initializeCache,
// reprojectFrame, depthCulling, fillGaps, directSamples, requestSamples, age, allocCacheItem), Camera.js, Raytracer.js,
// Sample.js, Pixel.js, Scene2.js and the three.js r139 the reference vendors (loaded from /root/reference).
// Math.random is replaced by a seeded xorshift32 (Driver.initializeCache draws pixel positions from it).  Per frame
// the camera state three.js is in (from, matrixWorld, matrixWorldInverse, projectionMatrix, projectionMatrixInverse) is
// recorded as INPUT; the colour frame, the sampling frame and the frame's statistics are the expected OUTPUT.
// Only data is written; no reference source enters the repository.
//
//   node oracle/gen_golden_rc_frames.mjs [/root/reference] [tests/golden/rc]
import fs from 'fs';
import os from 'os';
import path from 'path';
import { createRequire } from 'module';
import { fileURLToPath, pathToFileURL } from 'url';

const here = path.dirname(fileURLToPath(import.meta.url));
const refRoot = process.argv[2] || '/root/reference';
const outDir = path.resolve(process.argv[3] || path.join(here, '..', 'tests', 'golden', 'rc'));

function makeRng(seed) {
    let s = seed >>> 0 || 1;
    return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; };
}

async function main() {
    const require = createRequire(import.meta.url);
    global.THREE = require(path.join(refRoot, 'rendercache-js', 'js', 'three.js'));
    const tmp = fs.mkdtempSync(path.join(os.tmpdir(), 'ptr-rc-'));
    const realRandom = Math.random;
    try {
        const srcDir = path.join(refRoot, 'rendercache-js', 'src');
        fs.mkdirSync(path.join(tmp, 'src'));
        for (const f of ['Driver.js', 'Camera.js', 'Pixel.js', 'Raytracer.js', 'Sample.js', 'Color.js', 'Sphere.js', 'Material.js',
                         'Vector3.js', 'Scene.js', 'Scene2.js'])
            fs.copyFileSync(path.join(srcDir, f), path.join(tmp, 'src', f));
        fs.writeFileSync(path.join(tmp, 'package.json'), '{"type":"module"}');
        const imp = async n => (await import(pathToFileURL(path.join(tmp, 'src', n)).href)).default;
        const Driver = await imp('Driver.js'), Camera = await imp('Camera.js'), Raytracer = await imp('Raytracer.js');
        const Vector3 = await imp('Vector3.js'), Scene2 = await imp('Scene2.js');

        fs.mkdirSync(outDir, { recursive: true });
        const index = { generator: 'oracle/gen_golden_rc_frames.mjs', node: process.version, three: global.THREE.REVISION,
                        reference: 'rendercache-js/src Driver.nextFrame (unmodified), scene Scene2 (tests/golden/rc/index.json: rc_scene2)',
                        rng: 'xorshift32 (13, 17, 5), state = seed, value = state / 2^32, installed as Math.random',
                        layout: 'per frame, float64: from[3] matrixWorld[16] matrixWorldInverse[16] projectionMatrix[16] projectionMatrixInverse[16] | ' +
                                'threshold requestFactor candidates requests completeness totalPriority cacheUsage (threshold NaN = null);  ' +
                                '.u32 per frame: getColorFrame, getSamplingFrame (w*h each)', cases: {} };
        const cases = [
            { name: 'walk_96x64', w: 96, h: 64, ratio: 16, frames: 5, seed: 21,
              from: k => [0.5 * k, 0.25 * k, 0.4 * k], to: k => [0.2 * k, 0, -20] },
            { name: 'still_64x48', w: 64, h: 48, ratio: 8, frames: 4, seed: 22, from: () => [0, 1, 3], to: () => [0, 0, -20] },
            { name: 'turn_80x60', w: 80, h: 60, ratio: 4, frames: 4, seed: 23, from: k => [3 * k, 2, 6 - k], to: k => [-2 * k, 0, -20] },
            // Driver.nextFrame1SPP (:117-128, what index.js calls when its `raytrace` switch is on, with ratio 1): gap filling,
            // selection and trace only - no reprojection, no depth culling, no ageing
            { name: 'onespp_48x32', w: 48, h: 32, ratio: 1, frames: 4, seed: 24, method: 'nextFrame1SPP',
              from: k => [0.4 * k, 0.3, 2], to: k => [0, 0, -20] },
            { name: 'onespp_ratio3_60x40', w: 60, h: 40, ratio: 3, frames: 5, seed: 25, method: 'nextFrame1SPP',
              from: k => [0, 0.5 * k, 3 - 0.5 * k], to: k => [0.5 * k, 0, -20] },
            // odd width and height (the pixel the candidate loops leave behind depends on the height's parity, Driver.js:813-814)
            { name: 'odd_75x45', w: 75, h: 45, ratio: 4, frames: 5, seed: 26, from: k => [-0.6 * k, 0.5 + 0.2 * k, 2 + 0.3 * k], to: k => [0.4 * k, 0, -20] },
            // many frames at ratio 2 with a camera that swings back and forth: the cache's pages fill up and allocCacheItem has to
            // clear the oldest item of a page (:283-297)
            { name: 'long_64x40', w: 64, h: 40, ratio: 2, frames: 16, seed: 27,
              from: k => [4 * Math.sin(0.9 * k), 1 + Math.cos(0.7 * k), 4 + 2 * Math.cos(0.9 * k)], to: k => [3 * Math.sin(1.3 * k), 0, -20] },
            // Driver.cacheFactor (:32, read by createCache :245) set to 1 and to 0.5 between the constructor and prepare(): the cache
            // is as large as the frame / half of it, pages are full and allocCacheItem clears and reuses their oldest item
            { name: 'smallcache_64x40', w: 64, h: 40, ratio: 2, frames: 8, seed: 29, cacheFactor: 1,
              from: k => [3 * Math.sin(0.8 * k), 1, 4 + Math.cos(0.8 * k)], to: k => [2 * Math.sin(1.1 * k), 0, -20] },
            { name: 'halfcache_48x32', w: 48, h: 32, ratio: 1, frames: 5, seed: 30, cacheFactor: 0.5,
              from: k => [0.8 * k, 0.5, 3], to: k => [-0.5 * k, 0, -20] },
            // the driver's tunables (plain fields, Driver.js:28-42) away from their defaults
            { name: 'tuned_72x48', w: 72, h: 48, ratio: 4, frames: 6, seed: 31,
              tunables: { ageFactor: 2, interpolationZero: 5, interpolationRandom: 1, priorityMax: 100, depthThreshold: 60 },
              from: k => [1.5 * Math.sin(0.7 * k), 0.5 + 0.3 * k, 3], to: k => [0.6 * k, 0, -20] },
            { name: 'tuned_50x50', w: 50, h: 50, ratio: 3, frames: 5, seed: 32, tunables: { ageFactor: 3, priorityMax: 7, interpolationZero: 60, interpolationRandom: 0 },
              from: k => [-0.7 * k, 1, 2 + 0.5 * k], to: k => [0, 0.3 * k, -20] },
            { name: 'onespp_odd_45x27', w: 45, h: 27, ratio: 2, frames: 4, seed: 28, method: 'nextFrame1SPP',
              from: k => [0.3 * k, 0.2 * k, 2.5], to: k => [0, 0.2 * k, -20] },
        ];
        for (const cs of cases) {
            const scene = new Scene2();
            const camera = new Camera(new Vector3(...cs.from(0)), new Vector3(...cs.to(0)), 30, cs.w, cs.h);
            const tracer = new Raytracer(scene, camera);
            const drv = new Driver(tracer, camera, cs.ratio);
            if (cs.cacheFactor) drv.cacheFactor = cs.cacheFactor;
            drv.prepare(false);
            Object.assign(drv, cs.tunables || {});
            Math.random = makeRng(cs.seed);                     // after the constructors: three.js draws uuids from Math.random
            const rec = [], frames = new Uint32Array(2 * cs.w * cs.h * cs.frames);
            for (let k = 0; k < cs.frames; k++) {
                camera.updatePosition(new Vector3(...cs.from(k)), new Vector3(...cs.to(k)));
                camera.cam.updateMatrixWorld();                 // the state a page is in after three.js has rendered once
                rec.push(camera.from.x, camera.from.y, camera.from.z, ...camera.cam.matrixWorld.elements,
                         ...camera.cam.matrixWorldInverse.elements, ...camera.cam.projectionMatrix.elements,
                         ...camera.cam.projectionMatrixInverse.elements);
                drv[cs.method || 'nextFrame'](k);
                const st = drv.statistics;
                rec.push(st.threshold === null ? NaN : st.threshold, st.requestFactor, st.candidates, st.requests, st.completeness,
                         st.totalPriority, drv.getCacheUsage());
                drv.getColorFrame(frames.subarray((2 * k) * cs.w * cs.h, (2 * k + 1) * cs.w * cs.h));
                drv.getSamplingFrame(frames.subarray((2 * k + 1) * cs.w * cs.h, (2 * k + 2) * cs.w * cs.h));
                console.log(cs.name, 'frame', k, 'threshold', st.threshold, 'candidates', st.candidates, 'requests', st.requests,
                            'completeness', st.completeness.toFixed(2), 'usage', drv.getCacheUsage().toFixed(3));
            }
            fs.writeFileSync(path.join(outDir, `frames_${cs.name}.f64`), Buffer.from(new Float64Array(rec).buffer));
            fs.writeFileSync(path.join(outDir, `frames_${cs.name}.u32`), Buffer.from(frames.buffer));
            index.cases[cs.name] = { file: `frames_${cs.name}.f64`, frames_file: `frames_${cs.name}.u32`, width: cs.w, height: cs.h,
                                     ratio: cs.ratio, frames: cs.frames, seed: cs.seed, fov: 30, doubles_per_frame: 67 + 7,
                                     method: cs.method || 'nextFrame', cache_factor: cs.cacheFactor || 4,
                                     tunables: Object.assign({ ageFactor: 1, depthThreshold: 1e6, interpolationZero: 20, interpolationRandom: 5,
                                                               priorityMax: 255 }, cs.tunables || {}) };
        }
        fs.writeFileSync(path.join(outDir, 'frames_index.json'), JSON.stringify(index, null, 1));
    } finally {
        Math.random = realRandom;
        fs.rmSync ? fs.rmSync(tmp, { recursive: true, force: true }) : fs.rmdirSync(tmp, { recursive: true });
    }
}
main().catch(e => { console.error(e); process.exit(1); });
