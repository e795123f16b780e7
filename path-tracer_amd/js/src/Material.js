/**
 * Material — plain record (raytracer-js-master/src/Material.js:4-11).
 * Argument order is (surfaceColor, reflection, transparency, emissionColor).  An element whose
 * emissionColor has any channel > 0 is a light.  reflection / transparency are carried and
 * serialized but not used by the reference's shading (SURVEY.md F1).
 */
export default class Material {
    constructor(surfaceColor, reflection, transparency, emissionColor) {
        this.surfaceColor = surfaceColor;
        this.reflection = reflection;
        this.transparency = transparency;
        this.emissionColor = emissionColor;
    }
}
