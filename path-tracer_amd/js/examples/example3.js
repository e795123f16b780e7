// The banded renderer of raytracer-js-master/examples/example3.js:62-117: a RenderPlanner deals the frame out in
// `jobs` scanline bands, every band comes back through onUpdateReceived(sectionStart, sectionHeight, buf8), and when the
// planner has stopped running the pieces are put together (the reference: putImageData per piece), the scene is moved on,
// updateScene() + start() begin the next frame.
//   node examples/example3.js [--jobs 8] [--frames 60] [--width 640] [--height 480] [--threads] [--still] [--out last.ppm]
// --threads: the planner drives real worker_threads Workers with the reference's messages (one GPU context per Worker)
//            instead of in-process renderers;  --still: no animation (the frame of tests/golden/frames/example1_bands8_320x240.rgba
//            with --width 320 --height 240).
import { RenderPlanner } from '../index.js';
import { exampleScene, animate, sha256, writePPM, args } from './common.js';

const opt = args({ jobs: 8, frames: 60, width: 640, height: 480, step: 10, threads: false, still: false, out: '' });
const { scene, backgroundColor, moving } = exampleScene();

const frame8 = new Uint8ClampedArray(opt.width * opt.height * 4);          // stands in for the canvas
let bufferPieces = [];
let totalFrames = 0, startTime = 0;

const renderPlanner = new RenderPlanner(opt.jobs, scene, backgroundColor, opt.width, opt.height, { workerThreads: opt.threads });
renderPlanner.onError = (err) => { console.error(err); renderPlanner.dispose(); process.exit(1); };
renderPlanner.onUpdateReceived = function (sectionStart, sectionHeight, buf8) {
    bufferPieces.push({ buffer: buf8, start: sectionStart, height: sectionHeight });      // collect the pieces of one screen update
    if (renderPlanner.isRunning()) return;
    for (const piece of bufferPieces) frame8.set(piece.buffer, piece.start * opt.width * 4);
    bufferPieces = [];
    if (totalFrames === 0) startTime = Date.now();                          // (the first frame loads the GPU code)
    if (++totalFrames > opt.frames) return finish();
    setImmediate(() => {
        if (!opt.still) animate(moving, totalFrames * opt.step);            // update scene data
        renderPlanner.updateScene();
        renderPlanner.start();
    });
};

function finish() {
    const seconds = (Date.now() - startTime) / 1000;
    console.log(`Worker Count: ${opt.jobs}${opt.threads ? ' (worker_threads)' : ''} FPS = ${(opt.frames / seconds).toFixed(1)} Total Frames: ${opt.frames}`);
    console.log(`sha256 of the last frame ${sha256(frame8)}`);
    if (opt.out) writePPM(opt.out, frame8, opt.width, opt.height);
    renderPlanner.dispose();
}

renderPlanner.initialize();
renderPlanner.start();
