// Where a frame's time goes in the Node host path (GPU box): node test/time_host_path.mjs [width height reps]
import { RayTracer } from '../index.js';
import { exampleScene } from '../examples/common.js';
const w = Number(process.argv[2] || 1920), h = Number(process.argv[3] || 1080), reps = Number(process.argv[4] || 200);
const { scene, backgroundColor } = exampleScene();
const rt = new RayTracer(backgroundColor, scene);
const ms = (f) => { for (let i = 0; i < 5; i++) f(); const t = process.hrtime.bigint(); for (let i = 0; i < reps; i++) f(); return +(Number(process.hrtime.bigint() - t) / 1e6 / reps).toFixed(4); };
const nat = rt._native, hd = rt._handle;
const out = { size: `${w}x${h}` };
out.render_ms = ms(() => rt.render(w, h));
out.upload_only_ms = ms(() => rt._upload(false));
out.native_render_only_ms = ms(() => nat.render(hd, w, h, 0, h, nat.QUIRK_ROW_SHIFT));
out.arraybuffer_alloc_ms = ms(() => new ArrayBuffer(w * h * 4));
out.arraybuffer_alloc_touch_ms = ms(() => { const a = new Uint32Array(new ArrayBuffer(w * h * 4)); for (let i = 0; i < a.length; i += 1024) a[i] = 1; });
out.stats_ms = ms(() => rt.stats());
console.log(JSON.stringify(out));
rt.dispose();
