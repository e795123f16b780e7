// One frame of the example scene — what raytracer-js-master/examples/example1.js:48-75 does with a canvas, here with a
// file: build the scene through the reference's classes, `new RayTracer(backgroundColor, scene)`, one render() call.
//   node examples/example1.js [--width 640] [--height 480] [--mode hybrid|tolerant|exact|f32] [--out frame.ppm]
// The buffer has the reference's layout (its row quirk included: the image sits one row up, the last row is zero), so
// its SHA-256 can be compared with the unmodified reference's: 640 x 480 -> e2ece997...0be40e, 1920 x 1080 -> b2b9c765...8979b0.
import { RayTracer } from '../index.js';
import { exampleScene, sha256, writePPM, args } from './common.js';

const opt = args({ width: 640, height: 480, mode: 'hybrid', out: '' });
const REFERENCE_SHA256 = {
    '640x480': 'e2ece997c1509ecfa42733bd66d4e45d73ea202c296ecbf62472ec0a470be40e',
    '1920x1080': 'b2b9c765b7275e3d91f734fa80ff834670448cc4f44fadef02f035b5f88979b0',
    '3840x2160': '808f1c72d09fdb51e55a9651103526ff129927bac7bd48b359672028e5d7cabe',
};

const { scene, backgroundColor } = exampleScene();
const rayTracer = new RayTracer(backgroundColor, scene, undefined, { mode: opt.mode });

rayTracer.render(opt.width, opt.height);                       // (the first call also loads the GPU code)
const startTime = Date.now();
const buffer = rayTracer.render(opt.width, opt.height);
const totalDuration = (Date.now() - startTime) / 1000;

const buf8 = new Uint8ClampedArray(buffer);                    // what the reference hands to ImageData.data.set()
const digest = sha256(buf8);
const known = REFERENCE_SHA256[`${opt.width}x${opt.height}`];
console.log(`Render completed! Total duration = ${totalDuration} seconds, ${rayTracer.stats().retraced} pixels re-traced in float64`);
console.log(`sha256 ${digest}` + (known && opt.mode !== 'tolerant' && opt.mode !== 'f32'
    ? (digest === known ? '  (= the unmodified reference\'s frame)' : '  (DIFFERS from the unmodified reference\'s frame)') : ''));
if (opt.out) writePPM(opt.out, buf8, opt.width, opt.height);
rayTracer.dispose();
if (known && (opt.mode === 'hybrid' || opt.mode === 'exact') && digest !== known) process.exit(1);
