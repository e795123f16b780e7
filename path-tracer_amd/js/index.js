// Public surface: the reference's class names, same constructors and methods.
export { default as Vector3 } from './src/Vector3.js';
export { default as Material } from './src/Material.js';
export { default as Sphere } from './src/Sphere.js';
export { default as Plane } from './src/Plane.js';          // extension primitive (not in the reference)
export { default as Scene } from './src/Scene.js';
export { default as Color } from './src/Color.js';
export { default as Camera } from './src/Camera.js';
export { default as RayTracer } from './src/RayTracer.js';
export { default as RenderPlanner } from './src/RenderPlanner.js';
export { createRenderWorker } from './src/RenderWorker.js';   // the worker side of the reference's message protocol
export { default as RenderCache } from './src/RenderCache.js';
export { default as RenderCacheDriver, RcCamera } from './src/RenderCacheDriver.js';
