import { Worker } from 'worker_threads';
import RayTracer from './RayTracer.js';
import Scene from './Scene.js';
import Sphere from './Sphere.js';
import Vector3 from './Vector3.js';

/**
 * RenderPlanner — drop-in for raytracer-js-master/src/RenderPlanner.js:4-129.  The reference
 * spawns `jobCount` WebWorkers, sends each the serialized scene and one scanline band, and calls
 * onUpdateReceived(sectionStart, sectionHeight, buf8) as the bands come back.  Here a job is a band
 * rendered on the GPU (jobs are dealt round-robin over `options.devices`, default [0]); the
 * serialize -> deserialize hop of the worker protocol (Sphere.js:53-83, RenderWorker.js:27-46) is
 * kept so that updateScene() has the same snapshot semantics.
 *
 * options.workerThreads = true keeps the reference's shape to the letter instead: `jobCount` Node worker_threads Workers
 * running ./RenderWorker.js, each with its own addon instance and GPU context, driven by the reference's four messages
 * (elements / backgroundColor / dimensions / render -> result, RenderPlanner.js:61-109); dispose() closes them.
 *
 * Band maths as in the reference (:75-76): sectionHeight = floor(height / jobCount),
 * sectionStart = index * sectionHeight; rows past jobCount * sectionHeight are never rendered, and
 * each band carries the row quirk of RayTracer.render().
 */
export default class RenderPlanner {
    constructor(jobCount, scene, backgroundColor, width, height, options = {}) {
        this.jobCount = jobCount;
        this.scene = scene;
        this.backgroundColor = backgroundColor;
        this.width = width;
        this.height = height;
        this.running = false;
        this.completedJobs = 0;
        this.onUpdateReceived = function (sectionStart, sectionHeight, buf8) {};
        this.onError = function (err) { throw err; };
        this.serializeScene();

        const devices = options.devices || [0];
        this.workers = [];
        this.workerThreads = !!options.workerThreads;
        if (this.workerThreads) {
            for (let i = 0; i < jobCount; i++)
                this.workers.push(new Worker(new URL('./RenderWorker.js', import.meta.url),
                                             { workerData: { device: devices[i % devices.length], mode: options.mode } }));
            return;
        }
        for (let i = 0; i < jobCount; i++) {
            // one renderer ("worker") per job, with its private copy of the scene
            const w = { scene: new Scene(), background: new Vector3(0, 0, 0), dims: null };
            w.tracer = new RayTracer(w.background, w.scene, undefined, { device: devices[i % devices.length], mode: options.mode });
            this.workers.push(w);
        }
    }

    initialize() {
        for (let i = 0; i < this.workers.length; i++) this.prepareWorker(i, this.workers[i]);
    }

    start() {
        this.running = true;
        this.completedJobs = 0;
        for (let i = 0; i < this.workers.length; i++) this.startWorker(this.workers[i]);
    }

    serializeScene() {
        this.serializedElements = this.scene.getElements().map(el => el.serialize());
    }

    static _loadElements(worker, serialized) {
        worker.scene.clear();
        for (const e of serialized) worker.scene.add(Sphere.deserialize(e));
    }

    prepareWorker(index, worker) {
        if (this.workerThreads) return this._prepareThread(index, worker);
        RenderPlanner._loadElements(worker, this.serializedElements);
        worker.background.x = this.backgroundColor.x;
        worker.background.y = this.backgroundColor.y;
        worker.background.z = this.backgroundColor.z;
        const sectionHeight = Math.floor(this.height / this.jobCount);
        const sectionStart = Math.floor(index * sectionHeight);
        worker.dims = [this.width, this.height, sectionStart, sectionHeight];
    }

    // the reference's prepareWorker, message for message (RenderPlanner.js:61-101)
    _prepareThread(index, rendererWorker) {
        rendererWorker.postMessage({ action: 'elements', data: this.serializedElements });
        rendererWorker.postMessage({ action: 'backgroundColor', data: [this.backgroundColor.x, this.backgroundColor.y, this.backgroundColor.z] });
        const sectionHeight = Math.floor(this.height / this.jobCount);
        const sectionStart = Math.floor(index * sectionHeight);
        rendererWorker.postMessage({ action: 'dimensions', data: [this.width, this.height, sectionStart, sectionHeight] });
        rendererWorker.removeAllListeners('message');
        rendererWorker.removeAllListeners('error');
        rendererWorker.on('message', (m) => {
            if (m.action == 'result') {
                this.completedJobs++;
                if (this.completedJobs == this.jobCount) this.running = false;
                this.onUpdateReceived(sectionStart, sectionHeight, m.data);
            } else if (m.action == 'error') {
                this.running = false;
                this.onError(new Error(m.data));
            }
        });
        rendererWorker.on('error', (err) => { this.running = false; this.onError(err); });
    }

    startWorker(worker) {
        if (this.workerThreads) { worker.postMessage({ action: 'render' }); return; }
        const [w, h, sectionStart, sectionHeight] = worker.dims;
        worker.tracer.renderAsync(w, h, sectionStart, sectionHeight).then(buffer => {
            this.completedJobs++;
            if (this.completedJobs == this.jobCount) this.running = false;
            this.onUpdateReceived(sectionStart, sectionHeight, new Uint8ClampedArray(buffer));
        }).catch(err => {
            this.running = false;          // the reference stalls forever here; surface the error instead
            this.onError(err);
        });
    }

    /**
     * One whole frame (every row at its place, no row quirk) rendered on ALL the planner's GPUs at once: the frame is dealt
     * out in cyclic strips of 8 rows, one part per device, and the parts meet on the first device over RCCL (xGMI) before one
     * copy to the host - the collection that raytracer-js-master/examples/example3.js:66-86 does piece by piece with
     * putImageData.  Uses the scene snapshot of the last initialize() / updateScene().  Returns an ArrayBuffer.
     */
    renderFrameGathered(options = {}) {
        if (this.workerThreads) throw new Error('renderFrameGathered needs the in-process renderers (workerThreads: false)');
        const first = [];
        const seen = new Set();
        for (const w of this.workers) {              // one renderer per distinct GPU, in device order of appearance
            if (seen.has(w.tracer._device)) continue;
            seen.add(w.tracer._device);
            first.push(w);
        }
        for (const w of first) w.tracer._upload();
        const handles = first.map(w => w.tracer._handle);
        const native = first[0].tracer._native;
        if (!this._comm) { native.commInitAll(handles); this._comm = true; }
        return native.renderMulti(handles, this.width, this.height, first[0].tracer._modeFlags,
                                  options.maxDepth, options.spp, options.seed);
    }

    isRunning() { return this.running; }

    updateScene() {
        this.serializeScene();
        if (this.workerThreads) {
            for (const w of this.workers) w.postMessage({ action: 'elements', data: this.serializedElements });    // RenderPlanner.js:116-128
            return;
        }
        for (const w of this.workers) RenderPlanner._loadElements(w, this.serializedElements);
    }

    dispose() {
        if (this.workerThreads) { for (const w of this.workers) w.postMessage({ action: 'close' }); return; }
        for (const w of this.workers) w.tracer.dispose();
    }
}
