// TEST INFRASTRUCTURE — times the JavaScript CPU port under Node worker_threads, one Worker per scanline
// band (the reference's RenderPlanner split: band = floor(h / jobs) rows at index * band,
// RT/src/RenderPlanner.js:75-76).  Usage: node cpu_bench.mjs <threads> <seconds> [width height]
// Prints one JSON line: {"threads", "frames", "seconds", "rays", "mrays_per_s", "node"}.
import { Worker } from 'worker_threads';
import fs from 'fs';
import path from 'path';
import { fileURLToPath } from 'url';

const here = path.dirname(fileURLToPath(import.meta.url));
const scenes = JSON.parse(fs.readFileSync(path.join(here, '..', '..', 'tests', 'golden', 'scenes.json')));
const threads = parseInt(process.argv[2] || '1', 10), seconds = parseFloat(process.argv[3] || '5');
const width = parseInt(process.argv[4] || '1920', 10), height = parseInt(process.argv[5] || '1080', 10);
const sc = scenes.example1;
const band = Math.floor(height / threads);

function round(workers, frames) {
    return Promise.all(workers.map((w, i) => new Promise((resolve, reject) => {
        w.once('message', (m) => { w.removeListener('error', reject); resolve(m); });
        w.once('error', reject);
        w.postMessage({ action: 'render', frames, width, height, startY: i * band, scanHeight: i === threads - 1 ? height - i * band : band });
    })));
}

(async () => {
    const workers = [];
    for (let i = 0; i < threads; i++)
        workers.push(new Worker(path.join(here, 'cpu_worker.mjs'), { workerData: { elements: sc.elements, background: sc.background } }));
    await round(workers, 1);                                    // warm up (JIT)
    const t0 = process.hrtime.bigint();
    let frames = 0, rays = 0;
    for (;;) {
        const res = await round(workers, 1);
        frames++;
        rays += res.reduce((s, r) => s + r.rays, 0);
        if (Number(process.hrtime.bigint() - t0) / 1e9 >= seconds) break;
    }
    const dt = Number(process.hrtime.bigint() - t0) / 1e9;
    console.log(JSON.stringify({ threads, frames, seconds: dt, rays, mrays_per_s: rays / dt / 1e6, width, height, node: process.version }));
    for (const w of workers) w.postMessage({ action: 'exit' });
})().catch((e) => { console.error(e); process.exit(1); });
