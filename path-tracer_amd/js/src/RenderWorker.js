// RenderWorker — the worker side of the reference's message protocol (raytracer-js-master/src/RenderWorker.js:23-70) for
// Node worker_threads: a thread that loads the N-API addon for itself and renders its scanline band on the GPU.
//
// Messages in, `{action, data}` as the reference's RenderPlanner posts them (RenderPlanner.js:61-82, :103-109, :116-128):
//   {action: "elements", data: [serialized spheres]}      replace the worker's scene (Sphere.deserialize per element)
//   {action: "backgroundColor", data: [r, g, b]}
//   {action: "dimensions", data: [width, height, startY, scanHeight]}
//   {action: "render"}                                     -> {action: "result", data: Uint8ClampedArray} (RenderWorker.js:66-70)
// The band has the layout of RayTracer.render() (row quirk included), as in the reference.  The result's buffer is moved to
// the receiver (transfer list), not copied.  Two additions the reference lacks: {action: "close"} releases the GPU context
// and ends the thread's message loop, and a failing render posts {action: "error", data: message} instead of leaving the
// planner waiting forever (the reference's worker has no onerror path, SURVEY.md section 5).
//
// Use: `new Worker(new URL('./RenderWorker.js', import.meta.url), {workerData: {device: 0, mode: 'hybrid'}})`, or
// RenderPlanner's `{workerThreads: true}` option; createRenderWorker(post, options) is the handler itself, for a caller
// with a message channel of its own.
import { parentPort, workerData, isMainThread } from 'worker_threads';
import RayTracer from './RayTracer.js';
import Scene from './Scene.js';
import Sphere from './Sphere.js';
import Vector3 from './Vector3.js';

export function createRenderWorker(post, options = {}) {
    const scene = new Scene();
    const backgroundColor = new Vector3(0, 0, 0);
    let dims = [0, 0, 0, 0];
    let tracer = null;                       // created on the first render: a worker that never renders takes no GPU context

    return function onMessage(message) {
        // (a DOM-style event carries the message in .data; worker_threads delivers the message itself)
        const m = (message && message.action === undefined && message.data && message.data.action !== undefined) ? message.data : message;
        const action = m.action, data = m.data;
        if (action == 'elements') {
            scene.clear();
            for (let i = 0; i < data.length; i++) scene.add(Sphere.deserialize(data[i]));
        } else if (action == 'backgroundColor') {
            backgroundColor.x = data[0]; backgroundColor.y = data[1]; backgroundColor.z = data[2];
        } else if (action == 'dimensions') {
            dims = [data[0], data[1], data[2], data[3]];
        } else if (action == 'render') {
            try {
                if (!tracer) tracer = new RayTracer(backgroundColor, scene, undefined, options);
                const buffer = tracer.render(dims[0], dims[1], dims[2], dims[3]);
                post({ action: 'result', data: new Uint8ClampedArray(buffer) }, [buffer]);
            } catch (err) {
                post({ action: 'error', data: String(err && err.message || err) });
            }
        } else if (action == 'close') {
            if (tracer) { tracer.dispose(); tracer = null; }
            if (options.onClose) options.onClose();
        }
    };
}

if (!isMainThread && parentPort) {
    const opts = Object.assign({}, workerData || {});
    opts.onClose = () => parentPort.close();
    parentPort.on('message', createRenderWorker((msg, transfer) => parentPort.postMessage(msg, transfer), opts));
}
