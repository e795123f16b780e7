/**
 * Vector3 — mutable float64 3-vector with the reference's method set
 * (raytracer-js-master/src/Vector3.js:4-79): every operation works in place and returns `this`.
 */
export default class Vector3 {
    constructor(x = 0, y = 0, z = 0) {
        this.x = x;
        this.y = y;
        this.z = z;
    }

    clone() { return new Vector3(this.x, this.y, this.z); }

    length2() { return this.x * this.x + this.y * this.y + this.z * this.z; }

    length() { return Math.sqrt(this.length2()); }

    /** scales by 1/sqrt(len2) with three multiplies; a zero vector is left alone */
    normalize() {
        const len2 = this.length2();
        if (len2 > 0) {
            const inv = 1 / Math.sqrt(len2);
            this.x *= inv; this.y *= inv; this.z *= inv;
        }
        return this;
    }

    dotProduct(o) { return this.x * o.x + this.y * o.y + this.z * o.z; }

    /** component-wise product */
    product(o) { this.x *= o.x; this.y *= o.y; this.z *= o.z; return this; }

    multiply(s) { this.x *= s; this.y *= s; this.z *= s; return this; }

    add(o) { this.x += o.x; this.y += o.y; this.z += o.z; return this; }

    subtract(o) { this.x -= o.x; this.y -= o.y; this.z -= o.z; return this; }

    revert() { this.x = -this.x; this.y = -this.y; this.z = -this.z; return this; }

    /** copies THIS into `target` (argument is the destination, as in the reference) */
    copy(target) { target.x = this.x; target.y = this.y; target.z = this.z; return this; }
}
