// Host form against resident form of RenderCacheDriver at full size: node test/lockstep_rc_frames.mjs [w h ratio frames]
// Every array of the state, the statistics and the four frames are compared after every frame (a check, not a timing).
import fs from 'fs'; import path from 'path'; import { fileURLToPath } from 'url';
import { Vector3, Sphere, Scene, RayTracer, RenderCacheDriver, RcCamera } from '../index.js';
const here = path.dirname(fileURLToPath(import.meta.url));
const golden = path.join(here, '..', '..', '..', 'tests', 'golden');
const idx = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'frames_index.json')));
const desc = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'index.json'))).scenes.rc_scene2;
const m = idx.cases.walk_96x64;
const b = fs.readFileSync(path.join(golden, 'rc', m.file));
const rec = new Float64Array(b.buffer, b.byteOffset, b.length / 8);
const w = Number(process.argv[2] || 1920), h = Number(process.argv[3] || 1080), ratio = Number(process.argv[4] || 16), frames = Number(process.argv[5] || 12);
const make = resident => {
    const sc = new Scene(); for (const e of desc.elements) sc.add(Sphere.deserialize(e));
    const rt = new RayTracer(new Vector3(...desc.background), sc, undefined, { mode: 'exact' });
    const cam = new RcCamera(w, h), drv = new RenderCacheDriver(rt, cam, ratio, { resident });
    drv.prepare(false);
    return { rt, cam, drv };
};
const pair = [make(false), make(true)];
const arrays = ['itemPixel', 'itemHit', 'itemAge', 'itemRgb', 'pixelElement', 'pixelDepth', 'pixelRgb', 'pixelPriority', 'pixelSample'];
const px = w * h, fa = new Uint32Array(px), fb = new Uint32Array(px);
const realRandom = Math.random;
let bad = 0;
for (let k = 0; k < frames; k++) {
    const o = (k % m.frames) * m.doubles_per_frame;
    for (const d of pair) {
        let s = (4242 + k) >>> 0;
        Math.random = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; };
        d.cam.setView(rec.subarray(o, o + 3), rec.subarray(o + 3, o + 19), rec.subarray(o + 19, o + 35), rec.subarray(o + 35, o + 51), rec.subarray(o + 51, o + 67));
        d.drv.nextFrame(k);
    }
    Math.random = realRandom;
    const [host, gpu] = pair.map(d => d.drv);
    const diffs = [];
    if (JSON.stringify(host.statistics) !== JSON.stringify(gpu.statistics)) diffs.push('statistics');
    if (host.getCacheUsage() !== gpu.getCacheUsage()) diffs.push('cacheUsage');
    for (const f of ['getColorFrame', 'getReprojectionFrame', 'getPriorityFrame', 'getSamplingFrame']) {
        gpu[f](fa); host[f](fb);
        for (let i = 0; i < px; i++) if (fa[i] !== fb[i]) { diffs.push(f); break; }
    }
    gpu.syncToHost();
    for (const name of arrays) {
        const x = gpu[name], y = host[name];
        for (let i = 0; i < x.length; i++) if (!(x[i] === y[i] || (x[i] !== x[i] && y[i] !== y[i]))) { diffs.push(`${name}[${i}]`); break; }
    }
    console.log(`frame ${k}: requests ${gpu.statistics.requests}, threshold ${gpu.statistics.threshold}, ` + (diffs.length ? 'DIFFERS: ' + diffs.join(' ') : 'identical'));
    bad += diffs.length;
}
for (const d of pair) { d.drv.dispose(); d.rt.dispose(); }
console.log(bad ? 'FAILED' : `ok: ${frames} frames of ${w} x ${h} (ratio ${ratio}) identical in every array, statistic and frame`);
process.exit(bad ? 1 : 0);
