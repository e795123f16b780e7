// The render-cache loop of raytracer-js-master/examples/example4.js:52-90 (in full: rendercache-js/src/Driver.js): every
// frame the cached samples are reprojected into the new view, depth-culled, the gaps filled from the neighbours, and only
// width * height / ratio new rays are traced where the priorities are highest — `driver.prepare()` once, then
// `driver.nextFrame(frame)` + one of the frame getters per frame.  The camera here circles the scene slowly; three.js is
// not needed: RcCamera takes the four matrices a THREE.PerspectiveCamera would hold (common.js: perspectiveView).
//   node examples/example4.js [--frames 120] [--width 480] [--height 360] [--ratio 16] [--seed n] [--host] [--out last.ppm]
// --host: the loop's host form (the cache crosses PCIe every frame) instead of the cache resident in HBM.
import { RayTracer, RenderCacheDriver, RcCamera } from '../index.js';
import { exampleScene, perspectiveView, seedRandom, sha256, writePPM, args } from './common.js';

const opt = args({ frames: 120, width: 480, height: 360, ratio: 16, seed: 20240229, host: false, out: '' });
seedRandom(opt.seed);                                          // (the reference draws from the unseeded Math.random)
const { scene, backgroundColor } = exampleScene();
const raytracer = new RayTracer(backgroundColor, scene, undefined, { mode: 'exact' });    // the cache's rays are traced in float64
const camera = new RcCamera(opt.width, opt.height);
const driver = new RenderCacheDriver(raytracer, camera, opt.ratio, { resident: !opt.host });
driver.prepare(false);

const colorbuffer = new Uint32Array(opt.width * opt.height);
function look(frame) {
    const a = 0.002 * frame;                                   // a slow circle around the point the spheres stand on
    const v = perspectiveView([20 * Math.sin(a), 1 + 0.5 * Math.sin(3 * a), -20 + 20 * Math.cos(a)], [0, 0, -20], 30, opt.width / opt.height);
    camera.setView(v.from, v.matrixWorld, v.matrixWorldInverse, v.projectionMatrix, v.projectionMatrixInverse);
}

look(0);
driver.nextFrame(0);                                           // frame 0 fills a fifth of the cache at random positions
const startTime = Date.now();
for (let frame = 1; frame <= opt.frames; frame++) {
    look(frame);
    driver.nextFrame(frame);                                   // perform everything for the next frame
    driver.getColorFrame(colorbuffer);                         // compute frame data
}
const seconds = (Date.now() - startTime) / 1000;
const s = driver.statistics;
console.log(`FPS = ${(opt.frames / seconds).toFixed(1)} Total frame = ${opt.frames} (${opt.width} x ${opt.height}, 1 / ${opt.ratio} of the pixels ` +
            `traced per frame, cache ${opt.host ? 'on the host' : 'resident in HBM'})`);
console.log(`last frame: ${s.requests} rays requested, completeness ${Number(s.completeness).toFixed(4)}, cache usage ${driver.getCacheUsage().toFixed(4)}`);
console.log(`sha256 of the last colour frame ${sha256(colorbuffer)}`);
if (opt.out) {
    // the frame getters store rows bottom-up (Driver.js:965, what the reference's canvas transform expects): turn it over
    const rows = new Uint32Array(colorbuffer.length);
    for (let y = 0; y < opt.height; y++) rows.set(colorbuffer.subarray(y * opt.width, (y + 1) * opt.width), (opt.height - 1 - y) * opt.width);
    writePPM(opt.out, new Uint8Array(rows.buffer), opt.width, opt.height);
}
driver.dispose();
raytracer.dispose();
