// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// JavaScript CPU port of the per-pixel hot path of raytracer-js-master, written from the reference's
// behaviour (RayTracer.render RT/src/RayTracer.js:132-197, trace :18-95, Sphere.intersect RT/src/Sphere.js:15-33),
// in the reference's own style: mutable Vector3 objects, one allocation per intermediate vector.  It
// exists so that bench.py can time "the reference's JS path under Node worker_threads" on the GPU box's
// host cores, where the reference itself cannot be (cpu_baseline kind "port").  Pinned bit for bit by
// tests/golden/ (frames the unmodified reference produced): tests/test_oracle_js_port.py.
// Uses the product's API classes (Vector3, Sphere, Material); the product never imports this file.
import Vector3 from '../../path-tracer_amd/js/src/Vector3.js';
import Sphere from '../../path-tracer_amd/js/src/Sphere.js';

const FAR = 1e8, BIAS = 1e-4;

export function buildScene(serialized) { return serialized.map(e => Sphere.deserialize(e)); }

export const counters = { primary: 0, shadow: 0 };

// RayTracer.trace (RayTracer.js:18-95)
export function trace(elements, background, rayOrigin, rayDir) {
    let tnear = FAR, element = null;
    const hitInfo = { t0: FAR, t1: FAR };
    counters.primary++;
    for (let i = 0; i < elements.length; i++) {
        hitInfo.t0 = FAR; hitInfo.t1 = FAR;
        if (elements[i].intersect(rayOrigin, rayDir, hitInfo)) {
            if (hitInfo.t0 < 0) hitInfo.t0 = hitInfo.t1;
            if (hitInfo.t0 < tnear) { tnear = hitInfo.t0; element = elements[i]; }
        }
    }
    if (element === null) return background.clone();
    const surfaceColor = new Vector3(0, 0, 0);
    const hitPoint = rayOrigin.clone().add(rayDir.clone().multiply(tnear));
    const hitNormal = element.getNormal(hitPoint);
    if (rayDir.dotProduct(hitNormal) > 0) hitNormal.revert();
    const mat = element.getMaterial();
    const lightHit = { t0: FAR, t1: FAR };
    for (let i = 0; i < elements.length; i++) {
        const em = elements[i].getMaterial().emissionColor;
        if (em.x > 0 || em.y > 0 || em.z > 0) {
            counters.shadow++;
            const transmission = new Vector3(1, 1, 1);
            const lightDirection = elements[i].getCenter().clone().subtract(hitPoint).normalize();
            for (let j = 0; j < elements.length; j++) {
                if (i !== j) {
                    const shadowOrigin = hitPoint.clone().add(hitNormal.clone().multiply(BIAS));
                    if (elements[j].intersect(shadowOrigin, lightDirection, lightHit)) {
                        transmission.x = 0; transmission.y = 0; transmission.z = 0;
                        break;
                    }
                }
            }
            const ratio = Math.max(0, hitNormal.dotProduct(lightDirection));
            surfaceColor.add(mat.surfaceColor.clone().product(transmission).product(em.clone().multiply(ratio)));
        }
    }
    surfaceColor.add(mat.emissionColor);
    return surfaceColor;
}

// RayTracer.render (RayTracer.js:132-197); quirk = the reference's row indexing (RayTracer.js:169)
export function renderBand(elements, background, width, height, startY, scanHeight, quirk) {
    const buf = new ArrayBuffer(width * scanHeight * 4);
    const out = new Uint32Array(buf);
    const invW = 1 / width, invH = 1 / height, fov = 30, aspect = width / height;
    const angle = Math.tan(Math.PI * 0.5 * fov / 180);
    const origin = new Vector3(0, 0, 0);
    for (let y = startY; y < startY + scanHeight; y++) {
        for (let x = 0; x < width; x++) {
            const xx = (2 * ((x + 0.5) * invW) - 1) * angle * aspect;
            const yy = (1 - 2 * ((y + 0.5) * invH)) * angle;
            const dir = new Vector3(xx, yy, -1).normalize();
            const c = trace(elements, background, origin, dir);
            const r = Math.round(Math.min(1, c.x) * 255), g = Math.round(Math.min(1, c.y) * 255), b = Math.round(Math.min(1, c.z) * 255);
            const at = quirk ? width * (y - startY) - (width - x) : width * (y - startY) + x;
            if (at >= 0) out[at] = (255 << 24) | (b << 16) | (g << 8) | r;
        }
    }
    return buf;
}
