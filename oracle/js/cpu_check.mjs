// TEST INFRASTRUCTURE — pins the JavaScript CPU port to the golden frames of the unmodified reference.
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath } from 'url';
import Vector3 from '../../path-tracer_amd/js/src/Vector3.js';
import { buildScene, renderBand } from './cpu_port.mjs';

const here = path.dirname(fileURLToPath(import.meta.url));
const golden = path.join(here, '..', '..', 'tests', 'golden');
const index = JSON.parse(fs.readFileSync(path.join(golden, 'index.json')));
const scenes = JSON.parse(fs.readFileSync(path.join(golden, 'scenes.json')));
let checked = 0;
for (const [key, meta] of Object.entries(index.frames)) {
    const sc = scenes[key.split('@')[0]];
    const got = Buffer.from(renderBand(buildScene(sc.elements), new Vector3(...sc.background), meta.width, meta.height, 0, meta.height, true));
    const want = fs.readFileSync(path.join(golden, meta.file));
    if (Buffer.compare(got, want) !== 0) { console.error('MISMATCH ' + key); process.exit(1); }
    checked++;
}
const m = index.hashes['example1@640x480'], sc = scenes.example1;
const big = Buffer.from(renderBand(buildScene(sc.elements), new Vector3(...sc.background), m.width, m.height, 0, m.height, true));
if (crypto.createHash('sha256').update(big).digest('hex') !== m.sha256) { console.error('MISMATCH 640x480 sha256'); process.exit(1); }
console.log(`ok ${checked} golden frames + 640x480 sha256`);
