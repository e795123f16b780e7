/** Color — byte triple returned by RayTracer.requestPixel (raytracer-js-master/src/Color.js:3-8). */
export default class Color {
    constructor() { this.r = 0.0; this.g = 0.0; this.b = 0.0; }
}
