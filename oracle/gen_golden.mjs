// TEST INFRASTRUCTURE — golden-vector generator. Not part of the product.
//
// Runs the UNMODIFIED reference (raytracer-js-master/src/*.js, read from
// /root/reference) under Node and records its outputs as small fixtures under
// tests/golden/.  Only OUTPUTS (frames, hashes, per-ray float64 results) and the
// scene descriptions we invented for the probes are written; no reference source
// is copied into the repository.
//
// Recipe (SURVEY.md §8c): the reference files are ES modules without a package.json,
// so they are staged in a throw-away temp dir next to a {"type":"module"} package.json,
// imported from there, and the temp dir is deleted again.
//
//   node oracle/gen_golden.mjs [/root/reference] [tests/golden]
//
// Node 12: no top-level await -> everything lives in main().

import fs from 'fs';
import os from 'os';
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath, pathToFileURL } from 'url';

const here = path.dirname(fileURLToPath(import.meta.url));
const refRoot = process.argv[2] || '/root/reference';
const outDir = path.resolve(process.argv[3] || path.join(here, '..', 'tests', 'golden'));

// ---------------------------------------------------------------- PRNG (ours)
// xorshift32 -> doubles in [0,1); only used to invent probe scenes / rays.
function makeRng(seed) {
    let s = seed >>> 0 || 1;
    return function () {
        s ^= s << 13; s >>>= 0;
        s ^= s >>> 17;
        s ^= s << 5; s >>>= 0;
        return s / 4294967296;
    };
}

// ------------------------------------------------------- scene descriptions
// wire format = Sphere.serialize() of the reference (Sphere.js:53-69)
function S(c, r, sc, refl, transp, ec) {
    return { center: c, radius: r,
             material: { surfaceColor: sc, emissionColor: ec, transparency: transp, reflection: refl } };
}

// the 7-sphere list of raytracer-js-master/examples/example1.js:9-45
function example1Elements() {
    return [
        S([0.0, -10004, -20], 10000, [0.2, 0.2, 0.2], 0, 0, [0, 0, 0]),
        S([0, 0, -20], 4, [1.00, 0.32, 0.36], 1, 0.5, [0, 0, 0]),
        S([5, -1, -15], 2, [0.9, 0.76, 0.46], 1, 0, [0, 0, 0]),
        S([5, 0, -25], 3, [0.65, 0.77, 0.97], 1, 0, [0, 0, 0]),
        S([-5.5, 0, -15], 3, [0.9, 0.9, 0.9], 1, 0, [0, 0, 0]),
        S([0, 20, -30], 3, [0, 0, 0], 0, 0, [1.2, 1.2, 1.2]),
        S([0, 10, 10], 3, [0, 0, 0], 0, 0, [1, 1, 1]),
    ];
}

// example2.js:100-104 / example3.js:90-94 animation, at a fixed time t (ms)
function animatedElements(t) {
    const e = example1Elements();
    // example2.js:12-43 uses light1 at (0,10,-30) r3 em 1.2 — same list otherwise
    e[5].center = [10 * Math.sin(t / 2000), 10, -30];
    e[1].center = [0, 5 * Math.sin(t / 1000), -20];
    e[2].center = [5, -1 * Math.sin(t / 500), -15];
    e[3].center = [5, 6 * Math.cos(t / 1000), -25];
    e[4].center = [-5.5, 3 * Math.cos(t / 1000), -15];
    return e;
}

function buildScenes() {
    const scenes = {};
    scenes.example1 = { background: [2, 2, 2], elements: example1Elements() };
    scenes.anim_t1234 = { background: [2, 2, 2], elements: animatedElements(1234) };
    scenes.anim_t98765 = { background: [2, 2, 2], elements: animatedElements(98765) };

    // (6) variants from SURVEY §8c
    // light below the big sphere's horizon
    { const e = example1Elements(); e[5].center = [0, -30, -30]; e.pop();
      scenes.light_below_horizon = { background: [2, 2, 2], elements: e }; }
    // occluder BEHIND the light still blocks (F7: distance ignored)
    scenes.occluder_behind_light = { background: [0.1, 0.2, 0.3], elements: [
        S([0, -1004, -20], 1000, [0.5, 0.5, 0.5], 0, 0, [0, 0, 0]),
        S([0, 0, -20], 2, [0.9, 0.3, 0.3], 0, 0, [0, 0, 0]),
        S([0, 10, -20], 1, [0, 0, 0], 0, 0, [1.5, 1.4, 1.3]),
        S([0, 30, -20], 6, [0.3, 0.9, 0.3], 0, 0, [0, 0, 0]),   // behind the light, seen from below
    ] };
    // coincident spheres: first wins (strict <)
    scenes.coincident = { background: [0.3, 0.3, 0.3], elements: [
        S([0, 0, -10], 2, [1, 0, 0], 0, 0, [0, 0, 0]),
        S([0, 0, -10], 2, [0, 1, 0], 0, 0, [0, 0, 0]),
        S([4, 6, -4], 1, [0, 0, 0], 0, 0, [2, 2, 2]),
    ] };
    // no emission anywhere: black hits + background
    scenes.no_lights = { background: [0.25, 0.5, 0.75], elements: [
        S([0, 0, -10], 2, [1, 1, 1], 0, 0, [0, 0, 0]),
        S([3, 0, -12], 2, [1, 0, 1], 0, 0, [0, 0, 0]),
    ] };
    // scene order permuted (ties, accumulation order, first-blocker break)
    { const e = example1Elements(); const p = [6, 3, 0, 5, 1, 4, 2].map(i => e[i]);
      scenes.example1_permuted = { background: [2, 2, 2], elements: p }; }
    // camera inside a sphere (t0<0 -> t1), emissive surface that is also coloured,
    // single-channel emission, light that is itself visible
    scenes.inside_sphere = { background: [0, 0, 0], elements: [
        S([0, 0, 0], 50, [0.6, 0.7, 0.8], 0, 0, [0, 0, 0]),
        S([0, 0, -10], 2, [0.9, 0.2, 0.1], 0, 0, [0.05, 0, 0]),
        S([3, 4, -8], 0.5, [1, 1, 1], 0, 0, [0, 3, 0]),
        S([-6, 2, -12], 1, [0.2, 0.2, 0.2], 0, 0, [0.9, 0.9, 2.5]),
    ] };
    // empty scene: background only
    scenes.empty = { background: [0.2, 0.4, 0.6], elements: [] };
    // negative / >1 colours, NaN-free but exercising clamp and Math.round of negatives
    scenes.odd_colours = { background: [-0.2, 0.5, 7], elements: [
        S([0, 0, -10], 3, [-0.5, 2.0, 0.5], 0, 0, [0, 0, 0]),
        S([0, 8, -6], 1, [0, 0, 0], 0, 0, [1, 1, 1]),
    ] };

    // random scenes (ours): floor + k spheres + 1..3 lights
    const rng = makeRng(0xC0FFEE);
    const u = (a, b) => a + (b - a) * rng();
    for (let s = 0; s < 8; s++) {
        const n = [3, 5, 9, 16, 24, 33, 48, 64][s];
        const el = [];
        el.push(S([0, -1004, -20], 1000, [u(.1, .9), u(.1, .9), u(.1, .9)], 0, 0, [0, 0, 0]));
        const nl = 1 + (s % 3);
        for (let i = 1; i < n - nl; i++)
            el.push(S([u(-12, 12), u(-4, 8), u(-45, -8)], u(0.3, 2.5),
                      [u(.05, 1), u(.05, 1), u(.05, 1)], rng() < .5 ? 1 : 0, rng() < .3 ? .5 : 0, [0, 0, 0]));
        for (let i = 0; i < nl; i++)
            el.push(S([u(-20, 20), u(8, 30), u(-40, 15)], u(0.5, 3), [0, 0, 0], 0, 0,
                      [u(.3, 1.5), u(.3, 1.5), u(.3, 1.5)]));
        // shuffle so lights are not always last
        for (let i = el.length - 1; i > 0; i--) {
            const j = Math.floor(rng() * (i + 1)); const t = el[i]; el[i] = el[j]; el[j] = t;
        }
        scenes['random_' + s] = { background: [u(0, 2), u(0, 2), u(0, 2)], elements: el };
    }
    return scenes;
}

function sha256(buf) { return crypto.createHash('sha256').update(Buffer.from(buf)).digest('hex'); }

async function main() {
    // ---- stage the reference in a temp dir, import it, and clean up at the end
    const tmp = fs.mkdtempSync(path.join(os.tmpdir(), 'ptr-ref-'));
    try {
        const srcDir = path.join(refRoot, 'raytracer-js-master', 'src');
        fs.mkdirSync(path.join(tmp, 'src'));
        for (const f of fs.readdirSync(srcDir)) {
            if (f === 'Async.js') continue;              // does not parse; nothing imports it (SURVEY F9)
            fs.copyFileSync(path.join(srcDir, f), path.join(tmp, 'src', f));
        }
        fs.writeFileSync(path.join(tmp, 'package.json'), '{"type":"module"}');
        const imp = async n => (await import(pathToFileURL(path.join(tmp, 'src', n)).href)).default;
        const RayTracer = await imp('RayTracer.js');
        const Scene = await imp('Scene.js');
        const Sphere = await imp('Sphere.js');
        const Vector3 = await imp('Vector3.js');

        const mkScene = desc => {
            const sc = new Scene();
            for (const e of desc.elements) sc.add(Sphere.deserialize(e));
            return sc;
        };
        const mkTracer = desc => new RayTracer(new Vector3(...desc.background), mkScene(desc));

        fs.mkdirSync(path.join(outDir, 'frames'), { recursive: true });
        const scenes = buildScenes();
        fs.writeFileSync(path.join(outDir, 'scenes.json'), JSON.stringify(scenes, null, 1));

        const index = { generator: 'oracle/gen_golden.mjs', node: process.version,
                        reference: 'raytracer-js-master/src (unmodified)', frames: {}, hashes: {},
                        bands: {}, trace: {}, request_pixel: {}, intersect: {} };

        // ---- (1) small full frames, every scene, raw render() buffers (F5 row shift included)
        const smallSizes = { example1: [[64, 48], [160, 120], [33, 17]], default: [[64, 48]] };
        for (const name of Object.keys(scenes)) {
            for (const [w, h] of (smallSizes[name] || smallSizes.default)) {
                const buf = mkTracer(scenes[name]).render(w, h);
                const fn = `frames/${name}_${w}x${h}.rgba`;
                fs.writeFileSync(path.join(outDir, fn), Buffer.from(buf));
                index.frames[`${name}@${w}x${h}`] = { file: fn, width: w, height: h, sha256: sha256(buf) };
            }
        }

        // ---- (2) sha256 + probe pixels of the big frames
        const bigs = [['example1', 640, 480], ['example1', 1920, 1080], ['example1', 3840, 2160],
                      ['anim_t1234', 1920, 1080], ['random_5', 640, 480],
                      // hashes alone of every other scene at sizes the small frames do not reach (odd ones among them)
                      ...[0, 1, 2, 3, 4, 6, 7].map(s => ['random_' + s, 640, 480]),
                      ['inside_sphere', 640, 480], ['odd_colours', 640, 480], ['coincident', 640, 480],
                      ['light_below_horizon', 640, 480], ['occluder_behind_light', 640, 480], ['no_lights', 320, 240],
                      ['example1_permuted', 800, 600], ['anim_t98765', 1280, 720], ['example1', 1001, 563], ['random_7', 1279, 719]];
        for (const [name, w, h] of bigs) {
            const buf = mkTracer(scenes[name]).render(w, h);
            const v = new Uint32Array(buf);
            const probes = [];
            const prng = makeRng(w * 31 + h);
            for (let i = 0; i < 16; i++) {
                const x = i === 0 ? 0 : i === 1 ? w >> 1 : Math.floor(prng() * w);
                const y = i === 0 ? 0 : i === 1 ? h >> 1 : i === 2 ? h - 1 : Math.floor(prng() * h);
                probes.push([x, y, v[y * w + x]]);   // (x,y) index the STORED buffer (row shift included)
            }
            // row checksums: sum of u32 per stored row mod 2^32 (lets a test localise a mismatch)
            const rows = [];
            for (let y = 0; y < h; y++) { let s = 0; for (let x = 0; x < w; x++) s = (s + v[y * w + x]) >>> 0; rows.push(s); }
            index.hashes[`${name}@${w}x${h}`] = { width: w, height: h, sha256: sha256(buf), probes,
                                                   row_sums_sha256: sha256(new Uint32Array(rows).buffer) };
        }

        // ---- (3) banded render: the RenderPlanner split (RenderPlanner.js:75-76), 8 bands of 320x240
        {
            const w = 320, h = 240, jobs = 8, band = Math.floor(h / jobs);
            const parts = [];
            for (let i = 0; i < jobs; i++)
                parts.push(Buffer.from(mkTracer(scenes.example1).render(w, h, Math.floor(i * band), band)));
            fs.writeFileSync(path.join(outDir, 'frames/example1_bands8_320x240.rgba'), Buffer.concat(parts));
            index.bands['example1@320x240/8'] = { file: 'frames/example1_bands8_320x240.rgba', width: w, height: h,
                                                  jobs, band, sha256: sha256(Buffer.concat(parts)) };
            // uneven split: 7 jobs over 100 rows (trailing rows never rendered)
            const w2 = 48, h2 = 100, jobs2 = 7, band2 = Math.floor(h2 / jobs2);
            const parts2 = [];
            for (let i = 0; i < jobs2; i++)
                parts2.push(Buffer.from(mkTracer(scenes.random_3).render(w2, h2, Math.floor(i * band2), band2)));
            fs.writeFileSync(path.join(outDir, 'frames/random_3_bands7_48x100.rgba'), Buffer.concat(parts2));
            index.bands['random_3@48x100/7'] = { file: 'frames/random_3_bands7_48x100.rgba', width: w2, height: h2,
                                                 jobs: jobs2, band: band2, sha256: sha256(Buffer.concat(parts2)) };
        }

        // ---- (4) per-ray trace(o,d,0) float64 results; record = ox oy oz dx dy dz r g b
        const traceSets = {};
        const norm = (x, y, z) => { const l = Math.sqrt(x * x + y * y + z * z); return [x / l, y / l, z / l]; };
        {   // example1: lattice over the reference camera's frustum + special rays
            const rays = [];
            for (let j = 0; j < 24; j++) for (let i = 0; i < 32; i++) {
                const xx = (2 * ((i + 0.5) / 32) - 1) * 0.2679491924311227 * (4 / 3);
                const yy = (1 - 2 * ((j + 0.5) / 24)) * 0.2679491924311227;
                rays.push([0, 0, 0, ...norm(xx, yy, -1)]);
            }
            // horizon-grazing rays on the floor sphere: tangent elevation is where d2 == r2
            // tangent direction from origin to sphere C=(0,-10004,-20), r=1e4: sweep tiny offsets around it
            const C = [0, -10004, -20], r = 10000, L = Math.sqrt(C[1] * C[1] + C[2] * C[2]);
            const alpha = Math.asin(r / L);             // half-angle of the tangent cone
            const base = Math.atan2(C[1], -C[2]);       // elevation of the centre direction (negative)
            for (let k = -40; k <= 40; k++) {
                const el = base + alpha + k * 1e-9 * (Math.abs(k) > 20 ? 1000 : 1);
                for (const az of [0, 0.1, -0.2]) {
                    rays.push([0, 0, 0, ...norm(Math.sin(az) * Math.cos(el), Math.sin(el), -Math.cos(az) * Math.cos(el))]);
                }
            }
            // rays starting inside sphere 1, inside the floor sphere, behind things, towards lights
            rays.push([0, 0, -20, ...norm(0.3, 0.4, -1)]);
            rays.push([0, 1, -19, ...norm(-1, 0.2, 0.1)]);
            rays.push([0, -5, -20, ...norm(0, 1, 0)]);
            rays.push([0, -5, -20, ...norm(0, -1, 0)]);
            rays.push([0, 0, 0, ...norm(0, 20, -30)]);      // hits light 1
            rays.push([0, 0, 0, ...norm(0, 10, 10)]);       // hits light 2 (behind the camera)
            rays.push([0, 0, 0, ...norm(0, 1, 0)]);         // straight up: miss
            rays.push([0, 0, 0, ...norm(1, 0, 0)]);         // sideways: miss
            rays.push([0, 0, 0, 0, 0, -1]);                 // axis
            rays.push([100, 50, 30, ...norm(-100, -50, -50)]);
            const prng = makeRng(42);
            for (let i = 0; i < 400; i++) {
                const o = [prng() * 30 - 15, prng() * 12 - 3, prng() * 40 - 35];
                rays.push([...o, ...norm(prng() * 2 - 1, prng() * 2 - 1, prng() * 2 - 1)]);
            }
            traceSets.example1 = rays;
        }
        for (const name of ['inside_sphere', 'occluder_behind_light', 'coincident', 'random_4', 'odd_colours', 'empty']) {
            const rays = [];
            const prng = makeRng(name.length * 7919);
            for (let i = 0; i < 256; i++) {
                const o = i < 128 ? [0, 0, 0] : [prng() * 10 - 5, prng() * 6 - 2, prng() * 10 - 12];
                rays.push([...o, ...norm(prng() * 1.2 - 0.6, prng() * 0.9 - 0.45, -1)]);
            }
            traceSets[name] = rays;
        }
        for (const name of Object.keys(traceSets)) {
            const rays = traceSets[name];
            const rec = new Float64Array(rays.length * 9);
            const rt = mkTracer(scenes[name]);
            rays.forEach((q, k) => {
                const c = rt.trace(new Vector3(q[0], q[1], q[2]), new Vector3(q[3], q[4], q[5]), 0);
                rec.set(q, k * 9); rec[k * 9 + 6] = c.x; rec[k * 9 + 7] = c.y; rec[k * 9 + 8] = c.z;
            });
            const fn = `trace_${name}.f64`;
            fs.writeFileSync(path.join(outDir, fn), Buffer.from(rec.buffer));
            index.trace[name] = { file: fn, count: rays.length, record: 'ox oy oz dx dy dz r g b (float64 LE)' };
        }

        // ---- requestPixel(dir) -> bytes; record = dx dy dz r g b (float64)
        {
            const rays = traceSets.example1.filter(q => q[0] === 0 && q[1] === 0 && q[2] === 0);
            const rec = new Float64Array(rays.length * 6);
            rays.forEach((q, k) => {
                const c = mkTracer(scenes.example1).requestPixel(new Vector3(q[3], q[4], q[5]));
                rec.set([q[3], q[4], q[5], c.r, c.g, c.b], k * 6);
            });
            fs.writeFileSync(path.join(outDir, 'request_pixel_example1.f64'), Buffer.from(rec.buffer));
            index.request_pixel.example1 = { file: 'request_pixel_example1.f64', count: rays.length,
                                             record: 'dx dy dz r g b (float64 LE)' };
        }

        // ---- (5) Sphere.intersect table; record = ox oy oz dx dy dz cx cy cz radius hit t0 t1
        {
            const rows = [];
            const prng = makeRng(7);
            const cases = [];
            cases.push([[0, 0, 0], [0, 0, -1], [0, 0, -10], 2]);        // head-on
            cases.push([[0, 0, 0], [0, 0, 1], [0, 0, -10], 2]);         // behind: tca<0
            cases.push([[0, 0, 0], [0, 0, -1], [0, 5, -10], 2]);        // d2>r2
            cases.push([[0, 0, -10], [0, 0, -1], [0, 0, -10], 2]);      // at centre: tca=0, t0<0
            cases.push([[0, 0, -9], [0, 0, -1], [0, 0, -10], 2]);       // inside, centre ahead
            cases.push([[0, 0, -11], [0, 0, -1], [0, 0, -10], 2]);      // inside, centre behind -> false
            cases.push([[0, 2, 0], [0, 0, -1], [0, 0, -10], 2]);        // tangent
            for (let i = 0; i < 500; i++) {
                const o = [prng() * 20 - 10, prng() * 20 - 10, prng() * 20 - 10];
                let d = norm(prng() * 2 - 1, prng() * 2 - 1, prng() * 2 - 1);
                const c = [prng() * 20 - 10, prng() * 20 - 10, prng() * 20 - 10];
                const r = prng() < 0.1 ? 10000 * prng() : prng() * 6;
                if (i % 2) {   // aim near the sphere so that hits, grazes and near misses are all covered
                    const j = 1.6 * r;
                    d = norm(c[0] - o[0] + j * (prng() - 0.5), c[1] - o[1] + j * (prng() - 0.5), c[2] - o[2] + j * (prng() - 0.5));
                }
                cases.push([o, d, c, r]);
            }
            for (const [o, d, c, r] of cases) {
                const sp = Sphere.deserialize(S(c, r, [1, 1, 1], 0, 0, [0, 0, 0]));
                const out = { t0: 1e8, t1: 1e8 };
                const hit = sp.intersect(new Vector3(...o), new Vector3(...d), out);
                rows.push([...o, ...d, ...c, r, hit ? 1 : 0, out.t0, out.t1]);
            }
            const rec = new Float64Array(rows.length * 13);
            rows.forEach((q, k) => rec.set(q, k * 13));
            fs.writeFileSync(path.join(outDir, 'intersect_table.f64'), Buffer.from(rec.buffer));
            index.intersect = { file: 'intersect_table.f64', count: rows.length,
                                record: 'ox oy oz dx dy dz cx cy cz radius hit t0 t1 (float64 LE)' };
        }

        fs.writeFileSync(path.join(outDir, 'index.json'), JSON.stringify(index, null, 1));
        console.log('golden vectors written to', outDir);
        for (const k of Object.keys(index.hashes)) console.log(k, index.hashes[k].sha256);
    } finally {
        fs.rmSync ? fs.rmSync(tmp, { recursive: true, force: true }) : fs.rmdirSync(tmp, { recursive: true });
    }
}

main().catch(e => { console.error(e); process.exit(1); });
