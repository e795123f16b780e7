// Tests of the JavaScript host layer, written against the reference's API shape.
//   node test/run_tests.mjs cpu   -> host logic only (no GPU): classes, wire format, error paths
//   node test/run_tests.mjs gpu   -> renders through the N-API addon and compares with the golden
//                                    vectors the unmodified reference produced (tests/golden/)
import fs from 'fs';
import path from 'path';
import { fileURLToPath } from 'url';
import assert from 'assert';
import { Worker } from 'worker_threads';
import { spawnSync } from 'child_process';
import { Plane, Vector3, Material, Sphere, Scene, Color, Camera, RayTracer, RenderPlanner, RenderCache, RenderCacheDriver, RcCamera, createRenderWorker } from '../index.js';

const here = path.dirname(fileURLToPath(import.meta.url));
const golden = path.join(here, '..', '..', '..', 'tests', 'golden');
const index = JSON.parse(fs.readFileSync(path.join(golden, 'index.json')));
const scenes = JSON.parse(fs.readFileSync(path.join(golden, 'scenes.json')));
const mode = process.argv[2] || 'cpu';
let passed = 0;
function test(name, fn) { fn(); passed++; console.log('ok   ' + name); }
async function atest(name, fn) { await fn(); passed++; console.log('ok   ' + name); }

function f64(file) { const b = fs.readFileSync(path.join(golden, file)); return new Float64Array(b.buffer, b.byteOffset, b.length / 8); }
function u8(file) { return new Uint8Array(fs.readFileSync(path.join(golden, file))); }
function sameBytes(a, b) { return Buffer.compare(Buffer.from(a.buffer || a, a.byteOffset || 0, a.byteLength), Buffer.from(b.buffer || b, b.byteOffset || 0, b.byteLength)) === 0; }

// the scene-building lines of raytracer-js-master/examples/example1.js:9-45, unchanged
function buildExample1() {
    var scene = new Scene();
    scene.add(new Sphere(new Vector3(0.0, -10004, -20), 10000, new Material(new Vector3(0.2, 0.2, 0.2), 0, 0, new Vector3())));
    scene.add(new Sphere(new Vector3(0, 0, -20), 4, new Material(new Vector3(1.00, 0.32, 0.36), 1, 0.5, new Vector3())));
    scene.add(new Sphere(new Vector3(5, -1, -15), 2, new Material(new Vector3(0.9, 0.76, 0.46), 1, 0, new Vector3())));
    scene.add(new Sphere(new Vector3(5, 0, -25), 3, new Material(new Vector3(0.65, 0.77, 0.97), 1, 0, new Vector3())));
    scene.add(new Sphere(new Vector3(-5.5, 0, -15), 3, new Material(new Vector3(0.9, 0.9, 0.9), 1, 0, new Vector3())));
    scene.add(new Sphere(new Vector3(0, 20, -30), 3, new Material(new Vector3(), 0, 0, new Vector3(1.2, 1.2, 1.2))));
    scene.add(new Sphere(new Vector3(0, 10, 10), 3, new Material(new Vector3(), 0, 0, new Vector3(1, 1, 1))));
    return scene;
}
function sceneFrom(desc) { const s = new Scene(); for (const e of desc.elements) s.add(Sphere.deserialize(e)); return s; }

function cpuTests() {
    test('Vector3 in-place ops return this', () => {
        const v = new Vector3(1, 2, 3);
        assert.strictEqual(v.add(new Vector3(1, 1, 1)), v);
        assert.deepStrictEqual([v.x, v.y, v.z], [2, 3, 4]);
        assert.strictEqual(v.clone().multiply(2).subtract(v).dotProduct(new Vector3(1, 1, 1)), 9);
        assert.strictEqual(new Vector3().normalize().length2(), 0);           // zero vector untouched
        const n = new Vector3(3, 4, 0).normalize();
        assert.strictEqual(n.x, 3 * (1 / Math.sqrt(25)));
        const t = new Vector3(); new Vector3(7, 8, 9).copy(t); assert.deepStrictEqual([t.x, t.y, t.z], [7, 8, 9]);
        assert.deepStrictEqual(Object.keys(new Vector3()), ['x', 'y', 'z']);
        assert.strictEqual(new Vector3(1, -2, 3).revert().y, 2);
        assert.strictEqual(new Vector3(1, 2, 3).product(new Vector3(2, 3, 4)).z, 12);
    });
    test('Material argument order and field names', () => {
        const m = new Material(new Vector3(1, 2, 3), 0.25, 0.5, new Vector3(4, 5, 6));
        assert.strictEqual(m.reflection, 0.25); assert.strictEqual(m.transparency, 0.5);
        assert.strictEqual(m.surfaceColor.x, 1); assert.strictEqual(m.emissionColor.z, 6);
    });
    test('Sphere.intersect reproduces the reference table bit for bit', () => {
        const rec = f64(index.intersect.file);
        let hits = 0;
        for (let k = 0; k < index.intersect.count; k++) {
            const q = rec.subarray(k * 13, k * 13 + 13);
            const sp = new Sphere(new Vector3(q[6], q[7], q[8]), q[9], new Material(new Vector3(), 0, 0, new Vector3()));
            const out = { t0: 1e8, t1: 1e8 };
            const hit = sp.intersect(new Vector3(q[0], q[1], q[2]), new Vector3(q[3], q[4], q[5]), out);
            assert.strictEqual(hit ? 1 : 0, q[10]);
            assert.ok(Object.is(out.t0, q[11]) && Object.is(out.t1, q[12]));
            hits += hit;
        }
        assert.ok(hits > 50);
    });
    test('Sphere serialize / deserialize wire format', () => {
        const s = buildExample1().getElements()[1];
        const w = s.serialize();
        assert.deepStrictEqual(w, scenes.example1.elements[1]);
        const back = Sphere.deserialize(w);
        assert.deepStrictEqual(back.serialize(), w);
        assert.strictEqual(back.radius2, 16);
        assert.strictEqual(back.getNormal(new Vector3(0, 4, -20)).y, 1);
        assert.strictEqual(back.getCenter(), back.center); assert.strictEqual(back.getRadius(), 4);
    });
    test('Scene keeps order; flatten gives the C-ABI plane order', () => {
        const sc = buildExample1();
        assert.strictEqual(sc.getElements().length, 7);
        const soa = RayTracer.flatten(sc.getElements());
        assert.strictEqual(soa.length, 84);
        assert.strictEqual(soa[1 * 7 + 0], -10004);   // cy of element 0
        assert.strictEqual(soa[3 * 7 + 1], 4);        // radius of element 1
        assert.strictEqual(soa[5 * 7 + 1], 0.32);     // sg of element 1
        assert.strictEqual(soa[8 * 7 + 1], 0.5);      // transparency of element 1
        assert.strictEqual(soa[9 * 7 + 5], 1.2);      // er of element 5
        sc.clear(); assert.strictEqual(sc.getElements().length, 0);
    });
    test('Camera keeps its arguments (the reference mutates `to`)', () => {
        const to = new Vector3(0, 0, -1);
        const c = new Camera(new Vector3(0, 0, 0), to, 30, 640, 480);
        assert.strictEqual(c.angle, Math.tan(Math.PI * 0.5 * 30 / 180));
        assert.strictEqual(to.z, -1);
        const b = c.basis();
        assert.deepStrictEqual([b.right.x, b.up.y, b.forward.z], [1, 1, -1]);
        assert.ok(new Color().r === 0);
    });
    test('no CPU fallback: constructing a RayTracer without a GPU throws', () => {
        let threw = null;
        try { new RayTracer(new Vector3(2, 2, 2), buildExample1()); } catch (e) { threw = e; }
        if (process.env.PTR_EXPECT_GPU) return;
        assert.ok(threw, 'expected an error without a GPU');
        assert.ok(/no CPU fallback/.test(threw.message), threw.message);
    });
    test('worker protocol handler: scene, background and dimensions messages build the worker\'s state; render reports errors', () => {
        const posted = [];
        const onMessage = createRenderWorker((msg) => posted.push(msg), { device: 0 });
        const elements = buildExample1().getElements().map(el => el.serialize());
        onMessage({ action: 'elements', data: elements });
        onMessage({ data: { action: 'backgroundColor', data: [2, 2, 2] } });          // the DOM event shape is taken too
        onMessage({ action: 'dimensions', data: [320, 240, 30, 30] });
        onMessage({ action: 'elements', data: elements.slice(0, 3) });                 // replaces, does not append (RenderWorker.js:28)
        assert.strictEqual(posted.length, 0);
        if (!process.env.PTR_EXPECT_GPU) {
            onMessage({ action: 'render' });                                            // no GPU here: an error message, not a throw, not a hang
            assert.strictEqual(posted.length, 1);
            assert.strictEqual(posted[0].action, 'error');
            assert.ok(/no CPU fallback|HIP|device/i.test(posted[0].data));
        }
        onMessage({ action: 'close' });
    });
}

async function gpuTests() {
    test('example1.js scene lines + render(): bytes equal the reference frames', () => {
        const bg = new Vector3(2.0, 2.0, 2.0);
        const rt = new RayTracer(bg, buildExample1());
        for (const key of ['example1@64x48', 'example1@160x120', 'example1@33x17']) {
            const m = index.frames[key];
            const buf = rt.render(m.width, m.height);
            assert.ok(buf instanceof ArrayBuffer && buf.byteLength === m.width * m.height * 4);
            assert.ok(sameBytes(new Uint8Array(buf), u8(m.file)), key);
        }
        assert.deepStrictEqual([bg.x, bg.y, bg.z], [2, 2, 2]);          // user's object untouched
        assert.notStrictEqual(rt.render(8, 8), rt.render(8, 8));       // a fresh buffer each call
        rt.dispose();
    });
    test('all three precision modes on every golden scene', () => {
        for (const [key, m] of Object.entries(index.frames)) {
            const desc = scenes[key.split('@')[0]];
            for (const mode of ['hybrid', 'exact']) {
                const rt = new RayTracer(new Vector3(...desc.background), sceneFrom(desc), undefined, { mode });
                assert.ok(sameBytes(new Uint8Array(rt.render(m.width, m.height)), u8(m.file)), key + ' ' + mode);
                rt.dispose();
            }
        }
    });
    test('scene objects are re-read on every render (example2.js mutates centres)', () => {
        const scene = buildExample1();
        const rt = new RayTracer(new Vector3(2, 2, 2), scene);
        const a = Buffer.from(rt.render(64, 48));
        const t = 1234, els = scene.getElements();
        els[5].center = new Vector3(10 * Math.sin(t / 2000), 10, -30);
        els[1].center = new Vector3(0, 5 * Math.sin(t / 1000), -20);
        els[2].center = new Vector3(5, -1 * Math.sin(t / 500), -15);
        els[3].center = new Vector3(5, 6 * Math.cos(t / 1000), -25);
        els[4].center = new Vector3(-5.5, 3 * Math.cos(t / 1000), -15);
        const b = rt.render(64, 48);
        assert.ok(!sameBytes(a, new Uint8Array(b)));
        assert.ok(sameBytes(new Uint8Array(b), u8(index.frames['anim_t1234@64x48'].file)));
        rt.dispose();
    });
    test('trace() and requestPixel() against the reference per-ray vectors', () => {
        const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1());
        const rec = f64(index.trace.example1.file);
        const n = index.trace.example1.count;
        const o = new Float64Array(3 * n), d = new Float64Array(3 * n);
        for (let k = 0; k < n; k++) { o.set(rec.subarray(9 * k, 9 * k + 3), 3 * k); d.set(rec.subarray(9 * k + 3, 9 * k + 6), 3 * k); }
        const r = rt.traceBatch(o, d);
        for (let k = 0; k < n; k++) for (let c = 0; c < 3; c++) assert.ok(Object.is(r.rgb[3 * k + c], rec[9 * k + 6 + c]), `ray ${k}`);
        const one = rt.trace(new Vector3(rec[0], rec[1], rec[2]), new Vector3(rec[3], rec[4], rec[5]), 0);
        assert.ok(one instanceof Vector3 && Object.is(one.x, rec[6]));
        const rp = f64(index.request_pixel.example1.file);
        for (let k = 0; k < 40; k++) {
            const c = rt.requestPixel(new Vector3(rp[6 * k], rp[6 * k + 1], rp[6 * k + 2]));
            assert.deepStrictEqual([c.r, c.g, c.b], [rp[6 * k + 3], rp[6 * k + 4], rp[6 * k + 5]]);
        }
        const miss = rt.trace(new Vector3(0, 0, 0), new Vector3(0, 1, 0), 0);
        assert.deepStrictEqual([miss.x, miss.y, miss.z], [2, 2, 2]);   // unclamped background
        rt.dispose();
    });
    test('renderFrame(): every row at its true position', () => {
        const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1());
        const w = 64, h = 48;
        const shifted = new Uint32Array(rt.render(w, h)), plain = new Uint32Array(rt.renderFrame(w, h));
        for (let y = 1; y < h; y++) for (let x = 0; x < w; x++) assert.strictEqual(plain[y * w + x], shifted[(y - 1) * w + x]);
        assert.ok(plain.subarray((h - 1) * w).some(v => v !== 0));
        rt.dispose();
    });
    await atest('extension modes: renderFrame({maxDepth, spp, seed}) equals the frozen specification vectors', async () => {
        // NOT reference parity (the reference has no bounce loop and no sampling): tests/golden/ext holds
        // outputs of the float64 specification, which mode 'exact' reproduces bit for bit
        const ext = JSON.parse(fs.readFileSync(path.join(golden, 'ext', 'index.json'))).cases;
        const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1(), undefined, { mode: 'exact' });
        for (const key of ['example1@160x120_d5_s1_k0', 'example1@160x120_d5_s4_k7', 'example1@160x120_d0_s4_k7']) {
            const c = ext[key];
            const got = new Uint8Array(rt.renderFrame(c.width, c.height, { maxDepth: c.max_depth, spp: c.spp, seed: c.seed }));
            assert.ok(sameBytes(got, u8(c.file)), key);
        }
        const c = ext['example1@160x120_d5_s4_k7'];
        const viaPromise = new Uint8Array(await rt.renderFrameAsync(c.width, c.height, { maxDepth: 5, spp: 4, seed: 7 }));
        assert.ok(sameBytes(viaPromise, u8(c.file)));
        assert.ok(sameBytes(new Uint8Array(rt.renderFrame(64, 48, { maxDepth: 0, spp: 1 })), new Uint8Array(rt.renderFrame(64, 48))));
        assert.throws(() => rt.renderFrame(64, 48, { maxDepth: 17 }), RangeError);
        assert.throws(() => rt.renderFrame(64, 48, { spp: 0 }), RangeError);
        rt.dispose();
        const f32 = new RayTracer(new Vector3(2, 2, 2), buildExample1(), undefined, { mode: 'f32' });   // the float32 throughput form, by name
        const dflt = new RayTracer(new Vector3(2, 2, 2), buildExample1());                              // the default mode renders the specification
        assert.ok(sameBytes(new Uint8Array(dflt.renderFrame(c.width, c.height, { maxDepth: 5, spp: 4, seed: 7 })), u8(c.file)));
        dflt.dispose();
        const a = new Uint8Array(f32.renderFrame(c.width, c.height, { maxDepth: 5, spp: 4, seed: 7 })), b = u8(c.file);
        let off = 0;
        for (let i = 0; i < a.length; i++) if (Math.abs(a[i] - b[i]) > 1) off++;
        assert.ok(off <= 0.005 * a.length, `float32: ${off} bytes off by more than 1/255`);
        f32.dispose();
    });
    test('traceSamples(): the render cache\'s per-sample trace, bit for bit (rendercache-js goldens)', () => {
        const rc = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'index.json')));
        for (const name of ['rc_scene2', 'example1']) {
            const desc = rc.scenes[name];
            const rec = f64(path.join('rc', rc.samples[name].file));
            const rt = new RayTracer(new Vector3(...desc.background), sceneFrom(desc), undefined, { mode: 'exact' });
            const n = rec.length / 18;
            // rays are grouped by origin in the file
            let k0 = 0;
            while (k0 < n) {
                let k1 = k0;
                while (k1 < n && rec[18 * k1] === rec[18 * k0] && rec[18 * k1 + 1] === rec[18 * k0 + 1] && rec[18 * k1 + 2] === rec[18 * k0 + 2]) k1++;
                const dirs = new Float64Array(3 * (k1 - k0));
                for (let k = k0; k < k1; k++) for (let a = 0; a < 3; a++) dirs[3 * (k - k0) + a] = rec[18 * k + 3 + a];
                const out = rt.traceSamples(new Vector3(rec[18 * k0], rec[18 * k0 + 1], rec[18 * k0 + 2]), dirs);
                for (let k = k0; k < k1; k++) for (let a = 0; a < 3; a++) {
                    const j = 3 * (k - k0) + a;
                    assert.strictEqual(out.rgb[j], rec[18 * k + 6 + a]);
                    assert.strictEqual(out.hit[j], rec[18 * k + 9 + a]);
                    assert.strictEqual(out.normal[j], rec[18 * k + 12 + a]);
                    assert.strictEqual(out.rgb8[j], rec[18 * k + 15 + a]);
                }
                k0 = k1;
            }
            rt.dispose();
        }
    });
    test('RenderCache.reprojectDriver(): Driver.reprojectFrame of the render cache, bit for bit (rendercache-js goldens)', () => {
        const idx = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'reproject_index.json')));
        const rc = new RenderCache();
        for (const [name, m] of Object.entries(idx.cases)) {
            const rec = f64(path.join('rc', m.file));
            const w = m.width, h = m.height, n = m.items, stride = w + 2, cells = stride * (h + 2);
            // a Driver-shaped object built from the recorded inputs (same field names as the reference's)
            const buffer = [], addressY = [];
            for (let y = 0; y <= h + 1; y++) { addressY[y] = y * stride; for (let x = 0; x <= w + 1; x++) buffer.push({ x, y, element: null, depth: 1e13, sample: true }); }
            const cache = [];
            for (let i = 0; i < n; i++) {
                const o = 35 + 4 * i;
                cache.push({ hit: Number.isNaN(rec[o]) ? null : { x: rec[o], y: rec[o + 1], z: rec[o + 2] }, age: rec[o + 3], pixel: buffer[0] });
            }
            const driver = { cache, cacheSize: n, ageFactor: m.age_factor, depthThreshold: m.depth_threshold, buffer, addressY,
                             getPixel(x, y) { return this.buffer[this.addressY[y] + x]; },
                             camera: { from: { x: rec[0], y: rec[1], z: rec[2] }, scope: { x: w, y: h },
                                       cam: { matrixWorldInverse: { elements: Array.from(rec.subarray(3, 19)) },
                                              projectionMatrix: { elements: Array.from(rec.subarray(19, 35)) } } } };
            rc.reprojectDriver(driver);
            const o1 = 35 + 4 * n, o2 = o1 + 2 * n;
            for (let i = 0; i < n; i++) {
                assert.strictEqual(cache[i].age, rec[o1 + 2 * i], `${name} age ${i}`);
                const p = cache[i].pixel;
                assert.strictEqual(p ? p.y * stride + p.x : -1, rec[o1 + 2 * i + 1], `${name} pixel ${i}`);
            }
            for (let y = 1; y <= h; y++) for (let x = 1; x <= w; x++) {
                const px = driver.getPixel(x, y), c = y * stride + x;
                assert.strictEqual(px.element ? cache.indexOf(px.element) : -1, rec[o2 + 2 * c], `${name} element ${x},${y}`);
                assert.strictEqual(px.depth, rec[o2 + 2 * c + 1]);
                assert.strictEqual(px.sample, false);
            }
            assert.strictEqual(cells, (rec.length - o2) / 2);
        }
        rc.dispose();
    });
    test('RenderCache.shadeFrame(): Driver.depthCulling + fillGaps + frames, bit for bit (rendercache-js goldens)', () => {
        const idx = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'reproject_index.json')));
        const rc = new RenderCache();
        const eq = (x, y, msg) => assert.ok(x === y, `${msg || ''} ${x} !== ${y}`);      // Math.round gives -0 in the reference; an int32 holds 0
        for (const [name, m] of Object.entries(idx.cases)) {
            const rp = f64(path.join('rc', m.file)), sh = f64(path.join('rc', m.shade));
            const fb = fs.readFileSync(path.join(golden, 'rc', m.shade_frames));
            const frames = new Uint32Array(fb.buffer, fb.byteOffset, fb.length / 4);
            const w = m.width, h = m.height, n = m.items, stride = w + 2, cells = stride * (h + 2);
            const o1 = 35 + 4 * n, o2 = o1 + 2 * n;
            const pel = new Int32Array(cells), pdep = new Float64Array(cells), age = new Float64Array(n), rgb = new Uint32Array(n);
            for (let c = 0; c < cells; c++) { pel[c] = rp[o2 + 2 * c]; pdep[c] = rp[o2 + 2 * c + 1]; }
            for (let i = 0; i < n; i++) {
                age[i] = rp[o1 + 2 * i];
                if (!Number.isNaN(sh[3 * i])) rgb[i] = sh[3 * i] | (sh[3 * i + 1] << 8) | (sh[3 * i + 2] << 16);
            }
            const out = rc.shadeFrame(w, h, pel, pdep, age, rgb, { ageFactor: m.age_factor, interpolationZero: m.interpolation_zero,
                                      interpolationRandom: m.interpolation_random, priorityMax: m.priority_max });
            let o = 3 * n;
            for (let i = 0; i < n; i++) eq(age[i], sh[o + i], `${name} age ${i}`);
            o += n;
            for (let y = 1; y <= h; y++) for (let x = 1; x <= w; x++) {
                const c = y * stride + x, v = out.pixelRgb[c];
                eq(v & 255, sh[o + 4 * c]); eq((v >> 8) & 255, sh[o + 4 * c + 1]);
                eq((v >> 16) & 255, sh[o + 4 * c + 2]); eq(out.pixelPriority[c], sh[o + 4 * c + 3]);
            }
            o += 4 * cells;
            for (let k = 0; k < 768; k++) eq(out.hist[k], Number.isNaN(sh[o + k]) ? 0 : sh[o + k]);
            o += 768;
            for (let k = 0; k < 3; k++) eq(out.stats[k], sh[o + k]);
            const px = w * h;
            assert.ok(sameBytes(out.colorFrame, frames.subarray(0, px)) && sameBytes(out.reprojectionFrame, frames.subarray(px, 2 * px)) &&
                      sameBytes(out.priorityFrame, frames.subarray(2 * px, 3 * px)), name + ' frames');
        }
        rc.dispose();
    });
    for (const resident of [false, true])
    test(`RenderCacheDriver.nextFrame()${resident ? ', cache resident on the GPU' : ''}: whole render-cache frames, bit for bit (unmodified Driver.nextFrame${resident ? '' : ' / nextFrame1SPP'} goldens)`, () => {
        // tests/golden/rc/frames_*: rendercache-js Driver.nextFrame over several frames with a moving camera and a seeded
        // Math.random (the generator is named in frames_index.json); inputs = the camera state per frame, outputs = frames + statistics
        const idx = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'frames_index.json')));
        const desc = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'index.json'))).scenes.rc_scene2;
        const realRandom = Math.random;
        try {
            for (const [name, m] of Object.entries(idx.cases)) {
                const method = m.method || 'nextFrame';
                if (resident && method !== 'nextFrame') continue;            // (nextFrame1SPP is a host form)
                const rec = f64(path.join('rc', m.file));
                const fb = fs.readFileSync(path.join(golden, 'rc', m.frames_file));
                const frames = new Uint32Array(fb.buffer, fb.byteOffset, fb.length / 4);
                let s = m.seed >>> 0 || 1;                       // the generator's xorshift32
                Math.random = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; };
                const rt = new RayTracer(new Vector3(...desc.background), sceneFrom(desc), undefined, { mode: 'exact' });
                const cam = new RcCamera(m.width, m.height);
                const drv = new RenderCacheDriver(rt, cam, m.ratio, { resident });
                drv.cacheFactor = m.cache_factor;                // (4 unless the case says otherwise: a plain field, as in the reference)
                drv.prepare(false);
                Object.assign(drv, m.tunables);                  // (the defaults unless the case says otherwise)
                const px = m.width * m.height, got = new Uint32Array(px);
                for (let k = 0; k < m.frames; k++) {
                    const o = k * m.doubles_per_frame;
                    cam.setView(rec.subarray(o, o + 3), rec.subarray(o + 3, o + 19), rec.subarray(o + 19, o + 35),
                                rec.subarray(o + 35, o + 51), rec.subarray(o + 51, o + 67));
                    drv[method](k);
                    const st = drv.statistics, want = rec.subarray(o + 67, o + 74);
                    const tag = `${name} frame ${k}`;
                    assert.ok(Number.isNaN(want[0]) ? st.threshold === null : st.threshold === want[0], tag + ' threshold');
                    assert.strictEqual(st.requestFactor, want[1], tag + ' requestFactor');
                    assert.strictEqual(st.candidates, want[2], tag + ' candidates');
                    assert.strictEqual(st.requests, want[3], tag + ' requests');
                    assert.strictEqual(st.completeness, want[4], tag + ' completeness');
                    assert.strictEqual(st.totalPriority, want[5], tag + ' totalPriority');
                    assert.strictEqual(drv.getCacheUsage(), want[6], tag + ' cache usage');
                    assert.ok(sameBytes(drv.getColorFrame(got), frames.subarray(2 * k * px, (2 * k + 1) * px)), tag + ' colour frame');
                    assert.ok(sameBytes(drv.getSamplingFrame(got), frames.subarray((2 * k + 1) * px, (2 * k + 2) * px)), tag + ' sampling frame');
                }
                drv.dispose(); rt.dispose();
            }
        } finally {
            Math.random = realRandom;
        }
    });
    test('RenderCacheDriver: the resident frame loop leaves the same cache, grid and frames as the host walks (odd sizes, a frame that requests every pixel, a restart, other tunables, caches small enough that full pages reuse their oldest item)', () => {
        // no reference output exists at these sizes: the host form of the driver (bit-identical to the reference on the goldens
        // above) is the comparator; every array of the state is compared after every frame
        const idx = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'frames_index.json')));
        const desc = JSON.parse(fs.readFileSync(path.join(golden, 'rc', 'index.json'))).scenes.rc_scene2;
        const m = idx.cases.walk_96x64, rec = f64(path.join('rc', m.file));
        const realRandom = Math.random;
        const arrays = ['itemPixel', 'itemHit', 'itemAge', 'itemRgb', 'pixelElement', 'pixelDepth', 'pixelRgb', 'pixelPriority', 'pixelSample'];
        try {
            // cacheFactor (Driver.js:81-88 reads it in prepare) below 4: the cache is as large as the frame or smaller, its pages fill
            // up and allocCacheItem reuses the oldest item of a page (:283-297) - on the GPU rc_alloc's second branch
            for (const [w, h, ratio, nframes, tweak, before] of [[151, 97, 16, 7], [64, 33, 1, 4], [320, 200, 8, 6], [37, 2, 2, 5],
                    [96, 40, 8, 5, { ageFactor: 2, interpolationZero: 5, interpolationRandom: 1, priorityMax: 100, depthThreshold: 60 }],
                    [50, 50, 3, 4, { ageFactor: 3, priorityMax: 7 }],
                    [64, 40, 2, 10, null, { cacheFactor: 1 }], [48, 32, 1, 6, { ageFactor: 2 }, { cacheFactor: 0.5 }]]) {
                const drvs = [false, true].map(resident => {
                    const rt = new RayTracer(new Vector3(...desc.background), sceneFrom(desc), undefined, { mode: 'exact' });
                    const cam = new RcCamera(w, h), drv = new RenderCacheDriver(rt, cam, ratio, { resident });
                    Object.assign(drv, before || {});
                    drv.prepare(false);
                    Object.assign(drv, tweak || {});                                     // the reference's tunables are plain fields
                    return { rt, cam, drv };
                });
                const px = w * h, a = new Uint32Array(px), b = new Uint32Array(px);
                let reused = 0;                                                          // full pages met by the host walk
                const clearItem = drvs[0].drv._clearItem;
                drvs[0].drv._clearItem = function (i) { reused++; return clearItem.call(this, i); };
                for (let k = 0; k < nframes; k++) {
                    const frameIndex = (k === 4 && w === 151) ? 0 : k;                   // a restart in the middle of one sequence
                    const o = (k % m.frames) * m.doubles_per_frame, tag = `${w}x${h} ratio ${ratio} frame ${k}`;
                    for (const d of drvs) {
                        let s = (1000 + 17 * k) >>> 0;
                        Math.random = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; };
                        d.cam.setView(rec.subarray(o, o + 3), rec.subarray(o + 3, o + 19), rec.subarray(o + 19, o + 35),
                                      rec.subarray(o + 35, o + 51), rec.subarray(o + 51, o + 67));
                        d.drv.nextFrame(frameIndex);
                    }
                    const [host, gpu] = drvs.map(d => d.drv);
                    assert.deepStrictEqual(gpu.statistics, host.statistics, tag + ' statistics');
                    assert.strictEqual(gpu.getCacheUsage(), host.getCacheUsage(), tag + ' cache usage');
                    assert.strictEqual(gpu.totalPriority, host.totalPriority, tag + ' totalPriority');
                    for (const f of ['getColorFrame', 'getReprojectionFrame', 'getPriorityFrame', 'getSamplingFrame'])
                        assert.ok(sameBytes(gpu[f](a), host[f](b)), `${tag} ${f}`);
                    gpu.syncToHost();
                    const pages = Math.floor(host.cacheSize / 8);
                    assert.strictEqual((gpu.cachePointer / 8) % pages, (host.cachePointer / 8) % pages, tag + ' cachePointer');
                    for (const name of arrays) {
                        const x = gpu[name], y = host[name];
                        let bad = -1;
                        for (let i = 0; i < x.length && bad < 0; i++) if (!(x[i] === y[i] || (x[i] !== x[i] && y[i] !== y[i]))) bad = i;
                        assert.strictEqual(bad, -1, `${tag} ${name}[${bad}]: ${x[bad]} !== ${y[bad]}`);
                    }
                }
                if (before) assert.ok(reused > 100, `${w}x${h} cacheFactor ${before.cacheFactor}: only ${reused} items reused`);
                for (const d of drvs) { d.drv.dispose(); d.rt.dispose(); }
            }
        } finally {
            Math.random = realRandom;
        }
    });
    test('argument validation', () => {
        const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1());
        assert.throws(() => rt.render(0, 10), RangeError);
        assert.throws(() => rt.render(NaN, 10), TypeError);
        assert.throws(() => rt.render(65536, 65536), RangeError);
        assert.throws(() => new RayTracer(new Vector3(), new Scene(), undefined, { mode: 'nope' }), TypeError);
        const empty = new RayTracer(new Vector3(0.2, 0.4, 0.6), new Scene());
        assert.ok(sameBytes(new Uint8Array(empty.render(64, 48)), u8(index.frames['empty@64x48'].file)));
        rt.dispose(); empty.dispose();
        assert.throws(() => rt.render(8, 8), /disposed/);
    });
    await atest('RenderPlanner: 8 bands, worker protocol semantics, bytes equal the reference bands', async () => {
        const m = index.bands['example1@320x240/8'];
        const scene = buildExample1();
        const planner = new RenderPlanner(m.jobs, scene, new Vector3(2, 2, 2), m.width, m.height);
        const pieces = [];
        const done = new Promise((resolve, reject) => {
            planner.onError = reject;
            planner.onUpdateReceived = (start, height, buf8) => {
                assert.ok(buf8 instanceof Uint8ClampedArray);
                pieces.push({ start, height, buf8 });
                if (planner.isRunning() == false) resolve();
            };
        });
        planner.initialize();
        assert.strictEqual(planner.isRunning(), false);
        planner.start();
        assert.strictEqual(planner.isRunning(), true);
        await done;
        assert.strictEqual(pieces.length, m.jobs);
        pieces.sort((a, b) => a.start - b.start);
        const all = Buffer.concat(pieces.map(p => Buffer.from(p.buf8)));
        assert.ok(sameBytes(all, u8(m.file)));
        assert.deepStrictEqual(pieces.map(p => p.start), [0, 30, 60, 90, 120, 150, 180, 210]);
        // updateScene(): workers get a new snapshot; mutations after it are not seen until the next one
        scene.getElements()[1].center = new Vector3(0, 3, -20);
        planner.updateScene();
        scene.getElements()[1].center = new Vector3(0, -3, -20);
        const second = [];
        const done2 = new Promise((resolve, reject) => {
            planner.onError = reject;
            planner.onUpdateReceived = (start, height, buf8) => { second.push({ start, buf8 }); if (!planner.isRunning()) resolve(); };
        });
        planner.start();
        await done2;
        const ref = new RayTracer(new Vector3(2, 2, 2), (() => { const s = buildExample1(); s.getElements()[1].center = new Vector3(0, 3, -20); return s; })());
        second.sort((a, b) => a.start - b.start);
        assert.ok(sameBytes(Buffer.from(second[3].buf8), new Uint8Array(ref.render(320, 240, 90, 30))));
        ref.dispose(); planner.dispose();
    });
    await atest('worker protocol: 2 worker_threads Workers, each with its own addon instance, render the 8 bands through messages', async () => {
        // raytracer-js-master/src/RenderWorker.js:23-70 driven by hand, the way a caller with workers of its own would:
        // elements / backgroundColor / dimensions / render -> result; bytes = the unmodified reference's 8 bands
        const m = index.bands['example1@320x240/8'];
        const want = u8(m.file);
        const elements = buildExample1().getElements().map(el => el.serialize());
        const url = new URL('../src/RenderWorker.js', import.meta.url);
        const workers = [0, 1].map(() => new Worker(url, { workerData: { device: 0 } }));
        const exited = workers.map(w => new Promise(resolve => w.once('exit', resolve)));      // (listening from the start: a thread may be gone before the checks below are done)
        const band = Math.floor(m.height / m.jobs);
        const got = Buffer.alloc(want.length);
        await Promise.all(workers.map((w, k) => new Promise((resolve, reject) => {
            const mine = [];
            for (let j = k; j < m.jobs; j += 2) mine.push(j);          // worker k takes bands k, k + 2, ...
            let at = 0;
            const next = () => {
                w.postMessage({ action: 'dimensions', data: [m.width, m.height, mine[at] * band, band] });
                w.postMessage({ action: 'render' });
            };
            w.on('error', reject);
            w.on('message', (msg) => {
                if (msg.action == 'error') return reject(new Error(msg.data));
                assert.strictEqual(msg.action, 'result');
                assert.ok(msg.data instanceof Uint8ClampedArray && msg.data.length == m.width * band * 4);
                Buffer.from(msg.data.buffer).copy(got, mine[at] * band * m.width * 4);
                if (++at < mine.length) next(); else { w.postMessage({ action: 'close' }); resolve(); }
            });
            w.postMessage({ action: 'elements', data: elements });
            w.postMessage({ action: 'backgroundColor', data: [2, 2, 2] });
            next();
        })));
        assert.ok(sameBytes(got, want));
        await Promise.all(exited);                                                               // 'close' ends the threads
    });
    await atest('RenderPlanner {workerThreads: true}: the reference\'s planner over real Workers, bytes equal the reference bands', async () => {
        const m = index.bands['example1@320x240/8'];
        const scene = buildExample1();
        const planner = new RenderPlanner(m.jobs, scene, new Vector3(2, 2, 2), m.width, m.height, { workerThreads: true });
        const run = () => new Promise((resolve, reject) => {
            const pieces = [];
            planner.onError = reject;
            planner.onUpdateReceived = (start, height, buf8) => { pieces.push({ start, buf8 }); if (!planner.isRunning()) resolve(pieces); };
            planner.start();
        });
        planner.initialize();
        const pieces = await run();
        assert.strictEqual(pieces.length, m.jobs);
        pieces.sort((a, b) => a.start - b.start);
        assert.ok(sameBytes(Buffer.concat(pieces.map(p => Buffer.from(p.buf8))), u8(m.file)));
        scene.getElements()[1].center = new Vector3(0, 3, -20);
        planner.updateScene();                                          // a new snapshot goes to every worker
        scene.getElements()[1].center = new Vector3(0, -3, -20);
        const second = await run();
        const ref = new RayTracer(new Vector3(2, 2, 2), (() => { const s = buildExample1(); s.getElements()[1].center = new Vector3(0, 3, -20); return s; })());
        second.sort((a, b) => a.start - b.start);
        assert.ok(sameBytes(Buffer.from(second[3].buf8), new Uint8Array(ref.render(320, 240, 90, 30))));
        ref.dispose();
        const exits = planner.workers.map(w => new Promise(resolve => w.once('exit', resolve)));
        planner.dispose();
        await Promise.all(exits);
    });
    test('mode tolerant: every channel within 1 of the reference frame', () => {
        const m = index.frames['example1@160x120'];
        const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1(), undefined, { mode: 'tolerant' });
        const got = new Uint8Array(rt.render(m.width, m.height)), want = u8(m.file);
        let worst = 0;
        for (let k = 0; k < want.length; k++) worst = Math.max(worst, Math.abs(got[k] - want[k]));
        assert.ok(worst <= 1);
        rt.dispose();
    });
    await atest('dispose() right after renderAsync(): the job in flight finishes on a live context', async () => {
        const m = index.frames['example1@160x120'];
        for (let round = 0; round < 4; round++) {
            const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1());
            const pending = [rt.renderAsync(m.width, m.height), rt.renderAsync(m.width, m.height)];
            rt.dispose();                                     // the worker threads are (or will be) inside ptr_render
            assert.throws(() => rt.render(8, 8), /disposed/);
            for (const buf of await Promise.all(pending)) assert.ok(sameBytes(new Uint8Array(buf), u8(m.file)));
        }
    });
    test('RenderPlanner.renderFrameGathered(): cyclic strips over the planner\'s GPUs, gathered by the library (one GPU here)', () => {
        const scene = buildExample1();
        const planner = new RenderPlanner(4, scene, new Vector3(2, 2, 2), 200, 77, { devices: [0] });
        planner.initialize();
        const rt = new RayTracer(new Vector3(2, 2, 2), scene);
        const want = new Uint8Array(rt.renderFrame(200, 77));
        assert.ok(sameBytes(new Uint8Array(planner.renderFrameGathered()), want));
        scene.getElements()[1].center = new Vector3(0, 2, -20);
        planner.updateScene();
        assert.ok(sameBytes(new Uint8Array(planner.renderFrameGathered()), new Uint8Array(rt.renderFrame(200, 77))));
        rt.dispose(); planner.dispose();
    });
    test('Plane + Phong materials (extension): renderFrame equals the frozen specification frames', () => {
        // synthetic.planes_scene() of the Python side, built with the JS classes (tests/golden/ext: self-pinned, not reference parity)
        const ext = JSON.parse(fs.readFileSync(path.join(golden, 'ext', 'index.json'))).cases;
        const ex = scenes.example1.elements;
        const scene = new Scene();
        const ks = [0.6, 0.4, 0.5, 0.3, 0, 0], sh = [32, 37, 1, 128, 1, 1];
        // a plane added FIRST still counts after all spheres
        scene.add(new Plane(new Vector3(0, -4, 0), new Vector3(0, 2, 0), new Material(new Vector3(0.2, 0.2, 0.2), 0, 0, new Vector3(), 0.25, 8)));
        for (let i = 1; i < 7; i++) {
            const s = Sphere.deserialize(ex[i]);
            s.material = new Material(s.material.surfaceColor, s.material.reflection, s.material.transparency, s.material.emissionColor, ks[i - 1], sh[i - 1]);
            scene.add(s);
        }
        scene.add(new Plane(new Vector3(0, 0, -45), new Vector3(0, 0, 1), new Material(new Vector3(0.8, 0.8, 0.9), 1, 0, new Vector3())));
        scene.add(new Plane(new Vector3(30, 25, -20), new Vector3(-0.6, -0.8, 0), new Material(new Vector3(), 0, 0, new Vector3(0.3, 0.25, 0.2))));
        const rt = new RayTracer(new Vector3(2, 2, 2), scene);
        for (const key of ['planes_phong@160x120_d0_s1', 'planes_phong@160x120_d3_s2']) {
            const c = ext[key];
            const got = new Uint8Array(rt.renderFrame(c.width, c.height, { maxDepth: c.max_depth, spp: c.spp, seed: c.seed }));
            assert.ok(sameBytes(got, u8(c.file)), key);
        }
        assert.throws(() => rt.render(64, 48), RangeError);             // render() keeps the reference's semantics: no planes there
        const pl = scene.getElements()[0];
        const out = { t0: 1e8, t1: 1e8 };
        assert.ok(pl.intersect(new Vector3(0, 0, 0), new Vector3(0, -1, 0), out) && out.t0 === 4);
        assert.ok(!pl.intersect(new Vector3(0, 0, 0), new Vector3(0, 1, 0), out));
        assert.deepStrictEqual(Plane.deserialize(pl.serialize()).serialize(), pl.serialize());
        // back to a scene without extension data: the reference path again
        rt.scene = buildExample1();
        assert.ok(sameBytes(new Uint8Array(rt.render(64, 48)), u8(index.frames['example1@64x48'].file)));
        rt.dispose();
    });
    test('examples/: the four Node examples run as programs; example1-3 reproduce the reference\'s frames', () => {
        const run = (file, ...a) => {
            const r = spawnSync(process.execPath, [path.join(here, '..', 'examples', file), ...a], { encoding: 'utf8', timeout: 120000 });
            assert.strictEqual(r.status, 0, `${file} ${a.join(' ')}: ${r.stderr}${r.stdout}`);
            return r.stdout;
        };
        const sha = (out) => /sha256.*?\b([0-9a-f]{64})\b/.exec(out)[1];
        // example1: exits non-zero by itself when the frame is not the unmodified reference's
        assert.strictEqual(sha(run('example1.js')), index.hashes['example1@640x480'].sha256);
        assert.strictEqual(sha(run('example1.js', '--mode', 'exact', '--width', '1920', '--height', '1080')), index.hashes['example1@1920x1080'].sha256);
        assert.ok(/Render completed/.test(run('example1.js', '--mode', 'tolerant', '--width', '320', '--height', '200')));
        // example2: the animated scene at clock value 1234 is a golden frame of the reference
        assert.strictEqual(sha(run('example2.js', '--frames', '1', '--t0', '1234', '--width', '64', '--height', '48')), index.frames['anim_t1234@64x48'].sha256);
        assert.strictEqual(sha(run('example2.js', '--reuse', '--frames', '3', '--step', '0', '--t0', '1234', '--width', '64', '--height', '48')), index.frames['anim_t1234@64x48'].sha256);
        assert.ok(/FPS = /.test(run('example2.js', '--frames', '20', '--width', '320', '--height', '240')));
        // example3: 8 bands put together, in-process renderers and real Workers
        const bands = index.bands['example1@320x240/8'];
        assert.strictEqual(sha(run('example3.js', '--still', '--frames', '3', '--width', '320', '--height', '240')), bands.sha256);
        assert.strictEqual(sha(run('example3.js', '--still', '--threads', '--jobs', '8', '--frames', '2', '--width', '320', '--height', '240')), bands.sha256);
        assert.ok(/Total Frames: 10/.test(run('example3.js', '--frames', '10', '--width', '320', '--height', '240')));
        // example4: the render-cache loop, resident and host form leave the same frame (seeded Math.random)
        const a = run('example4.js', '--frames', '6', '--width', '96', '--height', '64', '--ratio', '8');
        const b = run('example4.js', '--frames', '6', '--width', '96', '--height', '64', '--ratio', '8', '--host');
        assert.strictEqual(sha(a), sha(b));
        assert.ok(/rays requested/.test(a));
    });
    test('RenderCacheDriver: random set-ups of the frame loop, resident against host walks (test/soak_rc_loop.mjs, a short run); a frame with no sample counted is refused by both', () => {
        const run = (...a) => {
            const r = spawnSync(process.execPath, [path.join(here, 'soak_rc_loop.mjs'), ...a], { encoding: 'utf8', timeout: 120000 });
            assert.strictEqual(r.status, 0, `soak_rc_loop ${a.join(' ')}: ${r.stderr}${r.stdout}`);
            return r.stdout;
        };
        assert.ok(/40 set-ups, \d+ frames .* equal after every frame/.test(run('40', '7')));
        // set-up 121 of seed 2 (130 x 18, priorityMax 7): its frame 7 counts no sample at any level - the reference's pick loop
        // (Driver.js:795-825) adds 0 to its index for ever; the host walk and ptr_rc_next_frame both refuse the frame instead
        assert.ok(/ 1 set-ups ended by a frame that both forms refuse/.test(run('122', '2', '121')));
    });
    test('renderInto(): the frame in the caller\'s own buffer equals render()\'s; wrong targets are refused', () => {
        const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1());
        const want = new Uint8Array(rt.render(160, 120));
        assert.ok(sameBytes(want, u8(index.frames['example1@160x120'].file)));
        const ab = new ArrayBuffer(160 * 120 * 4);
        assert.strictEqual(rt.renderInto(ab, 160, 120), ab);
        assert.ok(sameBytes(new Uint8Array(ab), want));
        const big = new Uint32Array(160 * 120 + 7).fill(0xdeadbeef);           // a typed array with room to spare: the rest stays
        rt.renderInto(big, 160, 120);
        assert.ok(sameBytes(new Uint8Array(big.buffer, 0, 160 * 120 * 4), want));
        assert.strictEqual(big[160 * 120], 0xdeadbeef);
        const view = new Uint8ClampedArray(new ArrayBuffer(160 * 120 * 4 + 16), 8, 160 * 120 * 4);    // an offset view
        rt.renderInto(view, 160, 120);
        assert.ok(sameBytes(view, want));
        const band = new Uint8Array(160 * 30 * 4);
        rt.renderInto(band, 160, 120, { startY: 30, scanHeight: 30 });
        assert.ok(sameBytes(band, new Uint8Array(rt.render(160, 120, 30, 30))));
        const whole = new Uint8Array(160 * 120 * 4);
        rt.renderInto(whole, 160, 120, { exact: false });
        assert.ok(sameBytes(whole, new Uint8Array(rt.renderFrame(160, 120))));
        assert.throws(() => rt.renderInto(new ArrayBuffer(100), 160, 120), RangeError);
        assert.throws(() => rt.renderInto(new Uint8Array(new ArrayBuffer(160 * 120 * 4 + 8), 1, 160 * 120 * 4), 160, 120), RangeError);
        assert.throws(() => rt.renderInto({}, 160, 120), TypeError);
        rt.dispose();
    });
    test('stats(): the reference ray mix at 640x480 (SURVEY 6)', () => {
        const rt = new RayTracer(new Vector3(2, 2, 2), buildExample1());
        rt.render(640, 480);
        const st = rt.stats();
        assert.ok(st.retraced > 0 && st.retraced < 0.05 * 640 * 480);
        rt.dispose();
    });
}

(async () => {
    if (mode === 'gpu') { process.env.PTR_EXPECT_GPU = '1'; cpuTests(); await gpuTests(); } else cpuTests();
    console.log(`${passed} JS tests passed (${mode})`);
})().catch(e => { console.error(e); process.exit(1); });
