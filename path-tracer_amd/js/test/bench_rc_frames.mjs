// Timing of whole render-cache frames (RenderCacheDriver.nextFrame) on the GPU box: node test/bench_rc_frames.mjs [w h ratio frames resident(0|1)]
// Camera states: the five of tests/golden/rc/frames_walk_96x64 (aspect 1.5), cycled.  Prints one JSON line.
import fs from 'fs'; import path from 'path'; import { fileURLToPath } from 'url';
import { Vector3, Sphere, Scene, RayTracer, RenderCacheDriver, RcCamera } from '../index.js';
const here = path.dirname(fileURLToPath(import.meta.url));
const golden = path.join(here, '..', '..', '..', 'tests', 'golden');
const idx = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'frames_index.json')));
const desc = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'index.json'))).scenes.rc_scene2;
const m = idx.cases.walk_96x64;
const b = fs.readFileSync(path.join(golden, 'rc', m.file));
const rec = new Float64Array(b.buffer, b.byteOffset, b.length / 8);
const w = Number(process.argv[2] || 480), h = Number(process.argv[3] || 320), ratio = Number(process.argv[4] || 16), frames = Number(process.argv[5] || 30), resident = Number(process.argv[6] || 0) !== 0;
const sc = new Scene(); for (const e of desc.elements) sc.add(Sphere.deserialize(e));
const rt = new RayTracer(new Vector3(...desc.background), sc, undefined, { mode: 'exact' });
const cam = new RcCamera(w, h), drv = new RenderCacheDriver(rt, cam, ratio, { resident });
drv.prepare(false);
const phases = resident ? { _residentFrame: 0 } : { reprojectFrame: 0, shadeFrame: 0, directSamples: 0, requestSamples: 0 };
for (const name of Object.keys(phases)) {
    const f = drv[name].bind(drv);
    drv[name] = (...a) => { const t = process.hrtime.bigint(); const r = f(...a); phases[name] += Number(process.hrtime.bigint() - t) / 1e6; return r; };
}
const setCam = k => { const o = (k % m.frames) * m.doubles_per_frame;
    cam.setView(rec.subarray(o, o + 3), rec.subarray(o + 3, o + 19), rec.subarray(o + 19, o + 35), rec.subarray(o + 35, o + 51), rec.subarray(o + 51, o + 67)); };
setCam(0); drv.nextFrame(0);                                       // frame 0 fills the cache
for (const k of Object.keys(phases)) phases[k] = 0;
const t0 = process.hrtime.bigint();
for (let k = 1; k <= frames; k++) { setCam(k); drv.nextFrame(k); }
const ms = Number(process.hrtime.bigint() - t0) / 1e6 / frames;
for (const k of Object.keys(phases)) phases[k] = +(phases[k] / frames).toFixed(3);
console.log(JSON.stringify({ what: 'RenderCacheDriver.nextFrame', width: w, height: h, ratio, frames, resident, ms_per_frame: +ms.toFixed(3), phases_ms: phases,
                             cache_items: drv.cacheSize, completeness: drv.statistics.completeness, requests: drv.statistics.requests,
                             cache_usage: drv.getCacheUsage() }));
drv.dispose(); rt.dispose();
