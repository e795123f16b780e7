// TEST INFRASTRUCTURE — golden vectors of the render cache's per-sample ray trace (SURVEY.md §8f rank 1).
//
// Runs the UNMODIFIED reference classes of rendercache-js (src/Raytracer.js:24-123 trace, src/Sample.js:49-75
// doRaytracing, Sphere, Vector3, Scene2) under Node and records, per ray: origin, direction, the unclamped colour
// trace() returns, sample.hit, sample.normalDir and the three byte values doRaytracing leaves in sample.color.
// Only outputs and the serialized scene descriptions are written; no reference source enters the repository.
// three.js is not needed: nothing on this path touches it (Camera.js, which does, is not imported).
//
//   node oracle/gen_golden_rc.mjs [/root/reference] [tests/golden/rc]
import fs from 'fs';
import os from 'os';
import path from 'path';
import { fileURLToPath, pathToFileURL } from 'url';

const here = path.dirname(fileURLToPath(import.meta.url));
const refRoot = process.argv[2] || '/root/reference';
const outDir = path.resolve(process.argv[3] || path.join(here, '..', 'tests', 'golden', 'rc'));

function makeRng(seed) {                       // xorshift32 (ours), only to invent probe rays
    let s = seed >>> 0 || 1;
    return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; };
}

async function main() {
    const tmp = fs.mkdtempSync(path.join(os.tmpdir(), 'ptr-rc-'));
    try {
        const srcDir = path.join(refRoot, 'rendercache-js', 'src');
        fs.mkdirSync(path.join(tmp, 'src'));
        for (const f of ['Raytracer.js', 'Sample.js', 'Color.js', 'Sphere.js', 'Material.js', 'Vector3.js', 'Scene.js', 'Scene2.js'])
            fs.copyFileSync(path.join(srcDir, f), path.join(tmp, 'src', f));
        fs.writeFileSync(path.join(tmp, 'package.json'), '{"type":"module"}');
        const imp = async n => (await import(pathToFileURL(path.join(tmp, 'src', n)).href)).default;
        const Raytracer = await imp('Raytracer.js'), Sample = await imp('Sample.js'), Sphere = await imp('Sphere.js');
        const Material = await imp('Material.js'), Vector3 = await imp('Vector3.js'), Scene = await imp('Scene.js');
        const Scene2 = await imp('Scene2.js');

        const serialize = sc => ({ background: [sc.backgroundColor.x, sc.backgroundColor.y, sc.backgroundColor.z],
                                   elements: sc.getElements().map(e => e.serialize()) });
        const main = JSON.parse(fs.readFileSync(path.join(here, '..', 'tests', 'golden', 'scenes.json')));
        const fromDesc = desc => {
            const sc = new Scene();
            for (const e of desc.elements) {
                const m = e.material;
                sc.add(new Sphere(new Vector3(...e.center), e.radius,
                    new Material(new Vector3(...m.surfaceColor), m.reflection, m.transparency, new Vector3(...m.emissionColor))));
            }
            sc.backgroundColor = new Vector3(...desc.background);
            return sc;
        };
        const scenes = { rc_scene2: new Scene2(), example1: fromDesc(main.example1), inside_sphere: fromDesc(main.inside_sphere),
                         random_4: fromDesc(main.random_4) };
        const origins = [[0, 0, 0], [0.5, 1.0, 3.0], [-8, 6, -12]];

        fs.mkdirSync(outDir, { recursive: true });
        const index = { generator: 'oracle/gen_golden_rc.mjs', node: process.version,
                        reference: 'rendercache-js/src Raytracer.trace + Sample.doRaytracing (unmodified)',
                        record: 'ox oy oz dx dy dz cr cg cb hx hy hz nx ny nz r g b (float64)', scenes: {}, samples: {} };
        for (const [name, sc] of Object.entries(scenes)) {
            index.scenes[name] = serialize(sc);
            const tracer = new Raytracer(sc, null);
            const rng = makeRng(0xC0DE + name.length);
            const rows = [];
            for (const o of origins) {
                const dirs = [];
                for (let iy = 0; iy < 12; iy++) for (let ix = 0; ix < 16; ix++)      // lattice over the view
                    dirs.push(new Vector3((ix + 0.5) / 16 - 0.5, 0.4 - (iy + 0.5) / 15, -1).normalize());
                const els = sc.getElements();
                for (let k = 0; k < 64 && els.length; k++) {                          // aimed at spheres, incl. grazing
                    const e = els[Math.floor(rng() * els.length)];
                    const a = 1 + 0.2 * (rng() - 0.5);
                    const t = new Vector3(e.center.x + a * e.radius * (2 * rng() - 1) - o[0],
                                          e.center.y + a * e.radius * (2 * rng() - 1) - o[1],
                                          e.center.z + a * e.radius * (2 * rng() - 1) - o[2]);
                    dirs.push(t.normalize());
                }
                dirs.push(new Vector3(0, 1, 0), new Vector3(0, -1, 0), new Vector3(0, 0, 1));
                for (const d of dirs) {
                    const from = new Vector3(...o);
                    const probe = new Sample(); probe.rayDir = d;
                    const c = tracer.trace(from, probe, 0);
                    const s = new Sample(); s.rayDir = d; s.pixel = { color: null };
                    s.doRaytracing(tracer, from, s);
                    rows.push(o[0], o[1], o[2], d.x, d.y, d.z, c.x, c.y, c.z, s.hit.x, s.hit.y, s.hit.z,
                              s.normalDir.x, s.normalDir.y, s.normalDir.z, s.color.r, s.color.g, s.color.b);
                }
            }
            const fn = `samples_${name}.f64`;
            fs.writeFileSync(path.join(outDir, fn), Buffer.from(new Float64Array(rows).buffer));
            index.samples[name] = { file: fn, rays: rows.length / 18 };
            console.log(name, rows.length / 18, 'rays');
        }
        fs.writeFileSync(path.join(outDir, 'index.json'), JSON.stringify(index, null, 1));
    } finally {
        fs.rmSync ? fs.rmSync(tmp, { recursive: true, force: true }) : fs.rmdirSync(tmp, { recursive: true });
    }
}
main().catch(e => { console.error(e); process.exit(1); });
