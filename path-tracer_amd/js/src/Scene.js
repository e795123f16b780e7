/**
 * Scene — ordered element list (raytracer-js-master/src/Scene.js:4-20).  Order is significant:
 * the nearest-hit loop keeps the first of equal distances and lights are accumulated in order.
 */
export default class Scene {
    constructor() { this.elements = []; }
    add(element) { this.elements.push(element); }
    getElements() { return this.elements; }
    clear() { this.elements = []; }
}
