// TEST INFRASTRUCTURE — golden vectors of the render cache's reprojection pass (SURVEY.md §8f rank 2).
//
// Runs the UNMODIFIED reference under Node: rendercache-js/src/Driver.js (reprojectFrame :417-486,
// detachFrameFromCache, prepare), src/Camera.js (reprojectPixel :52-74, updatePosition), src/Raytracer.js,
// src/Sample.js, src/Scene2.js and the three.js r139 the reference vendors (rendercache-js/js/three.js, loaded
// from /root/reference as the reference's own page does; it provides Vector3.project and the camera matrices).
// A cache is filled with hits traced from one camera position, then reprojected into a second camera; the
// inputs of that pass (hits, ages, the matrices three.js actually used) and its results (ages, item.pixel,
// pixel.element, pixel.depth) are recorded.  Only data is written; no reference source enters the repository.
//
//   node oracle/gen_golden_rc_reproject.mjs [/root/reference] [tests/golden/rc]
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
        const indexPath = path.join(outDir, 'reproject_index.json');
        const index = { generator: 'oracle/gen_golden_rc_reproject.mjs', node: process.version, three: global.THREE.REVISION,
                        reference: 'rendercache-js/src Driver.reprojectFrame + Camera.reprojectPixel (unmodified), three.js as vendored',
                        shade_layout: 'float64: per item colour r g b (NaN = none) | per item age after depthCulling | per pixel of the (w+2)x(h+2) grid: ' +
                                      'colour r g b, priority | 3 x 256 histogram bins (INTERPOLATED, URGENT, SAMPLED) | completeness, statistics.totalPriority, ' +
                                      'this.totalPriority;  .u32: getColorFrame, getReprojectionFrame, getPriorityFrame (w*h each)',
                        layout: 'float64: from[3] view[16] proj[16] | per item: hx hy hz (NaN = no hit) age_in | per item: age_out item_pixel | ' +
                                'per pixel of the (w+2)x(h+2) grid: element depth', cases: {} };
        const cases = [
            { name: 'move_96x64', w: 96, h: 64, fromA: [0, 0, 0], toA: [0, 0, -20], fromB: [1.5, 0.8, 2.0], toB: [0.5, 0, -20], seed: 11 },
            { name: 'turn_64x48', w: 64, h: 48, fromA: [0, 2, 5], toA: [0, 0, -20], fromB: [6, 3, 0], toB: [-3, 0, -22], seed: 12 },
            { name: 'same_48x32', w: 48, h: 32, fromA: [0, 0, 0], toA: [0, 0, -20], fromB: [0, 0, 0], toB: [0, 0, -20], seed: 13 },
            // odd sizes and the driver's tunables (plain fields, Driver.js:28-42) away from their defaults: a depth threshold that
            // rejects the far hits, an age factor that is no integer
            { name: 'tuned_45x27', w: 45, h: 27, fromA: [0, 1, 2], toA: [0, 0, -20], fromB: [2, 0.5, 4], toB: [-1, 0, -20], seed: 14,
              tunables: { ageFactor: 2.5, depthThreshold: 40, interpolationZero: 5, interpolationRandom: 1, priorityMax: 100 } },
        ];
        for (const cs of cases) {
            const rng = makeRng(cs.seed);
            const scene = new Scene2();
            const camera = new Camera(new Vector3(...cs.fromA), new Vector3(...cs.toA), 30, cs.w, cs.h);
            const tracer = new Raytracer(scene, camera);
            const drv = new Driver(tracer, camera, 16);
            drv.prepare(false);
            Object.assign(drv, cs.tunables || {});
            camera.updatePosition(new Vector3(...cs.fromA), new Vector3(...cs.toA));
            camera.cam.updateMatrixWorld();                    // camera A as the page would have it after a frame
            // fill about 70 % of the cache with hits traced through random pixels of camera A
            const n = drv.cacheSize;
            for (let i = 0; i < n; i++) {
                const item = drv.cache[i];
                item._i = i;                                   // tag for the record below (the reference never reads it)
                item.age = Math.floor(rng() * 30);
                if (rng() < 0.7) {
                    item.rayDir = camera.computeDirToXY(1 + rng() * (cs.w - 1), 1 + rng() * (cs.h - 1));
                    item.pixel = { color: null };             // doRaytracing stores the colour there too
                    item.doRaytracing(tracer, camera.from, item);   // sets item.hit / item.normalDir / item.color
                }
            }
            // special hits: exactly at the new eye (depth 0), behind the new camera, duplicates (equal depths)
            const Color = (await import(pathToFileURL(path.join(tmp, 'src', 'Color.js')).href)).default;
            const grey = () => { const c = new Color(); c.copy(128, 64, 32); return c; };
            drv.cache[3].hit = new Vector3(...cs.fromB); drv.cache[3].color = grey();
            drv.cache[5].hit = new Vector3(cs.fromB[0], cs.fromB[1], cs.fromB[2] + 50); drv.cache[5].color = grey();
            if (drv.cache[7].hit) for (const j of [9, 11, 400]) { drv.cache[j].hit = drv.cache[7].hit.clone(); drv.cache[j].color = grey(); }
            // camera B: the state three.js is in when reprojectFrame runs is what gets recorded
            camera.updatePosition(new Vector3(...cs.fromB), new Vector3(...cs.toB));
            camera.cam.updateMatrixWorld();
            const view = Array.from(camera.cam.matrixWorldInverse.elements), proj = Array.from(camera.cam.projectionMatrix.elements);
            const rec = [camera.from.x, camera.from.y, camera.from.z, ...view, ...proj];
            for (let i = 0; i < n; i++) {
                const it = drv.cache[i];
                if (it.hit) rec.push(it.hit.x, it.hit.y, it.hit.z); else rec.push(NaN, NaN, NaN);
                rec.push(it.age);
            }
            drv.reprojectFrame();
            const stride = cs.w + 2;
            for (let i = 0; i < n; i++) {
                const it = drv.cache[i];
                rec.push(it.age, it.pixel ? it.pixel.y * stride + it.pixel.x : -1);
            }
            for (let y = 0; y <= cs.h + 1; y++) for (let x = 0; x <= cs.w + 1; x++) {
                const p = drv.getPixel(x, y);
                rec.push(p.element ? p.element._i : -1, p.depth);
            }
            // ---- rank 3: depth culling + gap filling + the three frames, on the state reprojectFrame left
            const rec3 = [];
            for (let i = 0; i < n; i++) { const c = drv.cache[i].color; rec3.push(c ? c.r : NaN, c ? c.g : NaN, c ? c.b : NaN); }
            drv.totalPriority = 0;
            drv.depthCulling();
            drv.fillGaps();
            for (let i = 0; i < n; i++) rec3.push(drv.cache[i].age);
            for (let y = 0; y <= cs.h + 1; y++) for (let x = 0; x <= cs.w + 1; x++) {
                const p = drv.getPixel(x, y);
                rec3.push(p.color.r, p.color.g, p.color.b, p.priority);
            }
            for (let k = 0; k < 3; k++) for (let j = 0; j < 256; j++) rec3.push(drv.priorities[k][j]);
            rec3.push(drv.statistics.completeness, drv.statistics.totalPriority, drv.totalPriority);
            const frames = new Uint32Array(3 * cs.w * cs.h);
            drv.getColorFrame(frames.subarray(0, cs.w * cs.h));
            drv.getReprojectionFrame(frames.subarray(cs.w * cs.h, 2 * cs.w * cs.h));
            drv.getPriorityFrame(frames.subarray(2 * cs.w * cs.h));
            fs.writeFileSync(path.join(outDir, `shade_${cs.name}.f64`), Buffer.from(new Float64Array(rec3).buffer));
            fs.writeFileSync(path.join(outDir, `shade_${cs.name}.u32`), Buffer.from(frames.buffer));
            const oob = [0, 1, 2].map(k => Object.keys(drv.priorities[k]).filter(j => !(Number(j) >= 0 && Number(j) < 256)).length);
            console.log(cs.name, 'histogram keys outside 0..255:', oob, 'completeness', drv.statistics.completeness);
            const fn = `reproject_${cs.name}.f64`;
            fs.writeFileSync(path.join(outDir, fn), Buffer.from(new Float64Array(rec).buffer));
            index.cases[cs.name] = { file: fn, width: cs.w, height: cs.h, items: n, age_factor: drv.ageFactor, depth_threshold: drv.depthThreshold,
                                     shade: `shade_${cs.name}.f64`, shade_frames: `shade_${cs.name}.u32`,
                                     interpolation_zero: drv.interpolationZero, interpolation_random: drv.interpolationRandom, priority_max: drv.priorityMax };
            const attached = drv.cache.filter(it => it.pixel).length;
            console.log(cs.name, 'items', n, 'with pixel', attached);
        }
        fs.writeFileSync(indexPath, JSON.stringify(index, null, 1));
    } finally {
        fs.rmSync ? fs.rmSync(tmp, { recursive: true, force: true }) : fs.rmdirSync(tmp, { recursive: true });
    }
}
main().catch(e => { console.error(e); process.exit(1); });
