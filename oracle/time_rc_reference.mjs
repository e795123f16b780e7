// TEST INFRASTRUCTURE — times the UNMODIFIED rendercache-js Driver.nextFrame under Node (this container only: the
// reference does not travel), same configuration as path-tracer_amd/js/test/bench_rc_frames.mjs.
//   node oracle/time_rc_reference.mjs [w h ratio frames] [/root/reference]
import fs from 'fs'; import os from 'os'; import path from 'path';
import { createRequire } from 'module'; import { pathToFileURL } from 'url';
const w = Number(process.argv[2] || 480), h = Number(process.argv[3] || 320), ratio = Number(process.argv[4] || 16), frames = Number(process.argv[5] || 30);
const refRoot = process.argv[6] || '/root/reference';
async function main() {
    const require = createRequire(import.meta.url);
    global.THREE = require(path.join(refRoot, 'rendercache-js', 'js', 'three.js'));
    const tmp = fs.mkdtempSync(path.join(os.tmpdir(), 'ptr-rc-'));
    try {
        fs.mkdirSync(path.join(tmp, 'src'));
        for (const f of ['Driver.js', 'Camera.js', 'Pixel.js', 'Raytracer.js', 'Sample.js', 'Color.js', 'Sphere.js', 'Material.js', 'Vector3.js', 'Scene.js', 'Scene2.js'])
            fs.copyFileSync(path.join(refRoot, 'rendercache-js', 'src', f), path.join(tmp, 'src', f));
        fs.writeFileSync(path.join(tmp, 'package.json'), '{"type":"module"}');
        const imp = async n => (await import(pathToFileURL(path.join(tmp, 'src', n)).href)).default;
        const Driver = await imp('Driver.js'), Camera = await imp('Camera.js'), Raytracer = await imp('Raytracer.js');
        const Vector3 = await imp('Vector3.js'), Scene2 = await imp('Scene2.js');
        const from = k => [0.5 * (k % 5), 0.25 * (k % 5), 0.4 * (k % 5)], to = k => [0.2 * (k % 5), 0, -20];
        const camera = new Camera(new Vector3(...from(0)), new Vector3(...to(0)), 30, w, h);
        const drv = new Driver(new Raytracer(new Scene2(), camera), camera, ratio);
        drv.prepare(false);
        const step = k => { camera.updatePosition(new Vector3(...from(k)), new Vector3(...to(k))); camera.cam.updateMatrixWorld(); drv.nextFrame(k); };
        step(0);
        const t0 = process.hrtime.bigint();
        for (let k = 1; k <= frames; k++) step(k);
        const ms = Number(process.hrtime.bigint() - t0) / 1e6 / frames;
        console.log(JSON.stringify({ what: 'rendercache-js Driver.nextFrame (unmodified reference, Node ' + process.version + ', one thread)', width: w, height: h,
                                     ratio, frames, ms_per_frame: +ms.toFixed(3), completeness: drv.statistics.completeness }));
    } finally { fs.rmSync ? fs.rmSync(tmp, { recursive: true, force: true }) : fs.rmdirSync(tmp, { recursive: true }); }
}
main().catch(e => { console.error(e); process.exit(1); });
