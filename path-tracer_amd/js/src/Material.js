/**
 * Material — plain record (raytracer-js-master/src/Material.js:4-11).
 * Argument order is (surfaceColor, reflection, transparency, emissionColor).  An element whose
 * emissionColor has any channel > 0 is a light.  reflection / transparency are carried and
 * serialized but not used by the reference's shading (SURVEY.md F1).
 * EXTENSION (not in the reference): optional 5th / 6th arguments specular (Phong ks >= 0) and shininess (an integer
 * exponent 1..65536); a scene in which any material has specular > 0 is rendered with Phong terms (include/ptr.h).
 */
export default class Material {
    constructor(surfaceColor, reflection, transparency, emissionColor, specular, shininess) {
        this.surfaceColor = surfaceColor;
        this.reflection = reflection;
        this.transparency = transparency;
        this.emissionColor = emissionColor;
        if (specular !== undefined) { this.specular = specular; this.shininess = shininess === undefined ? 1 : shininess; }
    }
}
