// The animation of raytracer-js-master/examples/example2.js:66-113: every frame the light and four spheres get new centres
// (assigned, `sphere.center = new Vector3(...)`: the renderer re-reads the scene objects on every call) and the whole
// frame is rendered again.  The browser's setInterval / Date.now() become a loop over a clock that advances by --step
// milliseconds per frame, so a run is reproducible.
//   node examples/example2.js [--frames 300] [--width 640] [--height 480] [--t0 0] [--step 10] [--mode hybrid] [--reuse] [--out last.ppm]
// --reuse: every frame into ONE buffer (RayTracer.renderInto) instead of a new ArrayBuffer per frame, which is what the
// reference's API hands out and what costs most of a frame's time here.
// `--frames 1 --t0 1234 --width 64 --height 48` is the frame tests/golden/frames/anim_t1234_64x48.rgba holds.
import { RayTracer } from '../index.js';
import { exampleScene, animate, sha256, writePPM, args } from './common.js';

const opt = args({ frames: 300, width: 640, height: 480, t0: 0, step: 10, mode: 'hybrid', reuse: false, out: '' });
const { scene, backgroundColor, moving } = exampleScene();
const rayTracer = new RayTracer(backgroundColor, scene, undefined, { mode: opt.mode });

let isFirstFrame = 0, buffer = null, reTraced = 0;
function render() {
    if (opt.reuse) buffer = rayTracer.renderInto(buffer || new ArrayBuffer(opt.width * opt.height * 4), opt.width, opt.height);
    else buffer = rayTracer.render(opt.width, opt.height, undefined, undefined, isFirstFrame);
    isFirstFrame = 1;
    reTraced += rayTracer.stats().retraced;
    return new Uint8ClampedArray(buffer);
}

animate(moving, opt.t0);
render();                                                       // (loads the GPU code; not timed)
const startTime = Date.now();
let buf8 = null;
for (let frame = 0; frame < opt.frames; frame++) {
    animate(moving, opt.t0 + frame * opt.step);                 // update positions
    buf8 = render();                                            // render
}
const seconds = (Date.now() - startTime) / 1000;
console.log(`FPS = ${(opt.frames / seconds).toFixed(1)} Total frame = ${opt.frames} (${opt.width} x ${opt.height}, ` +
            `${(seconds / opt.frames * 1e3).toFixed(3)} ms per frame with scene upload and read-back)`);
console.log(`sha256 of the last frame ${sha256(buf8)}`);
if (opt.out) writePPM(opt.out, buf8, opt.width, opt.height);
rayTracer.dispose();
