// Loads the N-API addon (addon/ptr_napi.node -> libptr_hip.so).  There is no JavaScript or CPU
// implementation of the render path behind it: if the addon or a GPU is missing, this throws.
import { createRequire } from 'module';
import { fileURLToPath } from 'url';
import path from 'path';

const require = createRequire(import.meta.url);
const here = path.dirname(fileURLToPath(import.meta.url));

let addon = null;
export default function native() {
    if (!addon) {
        const file = path.join(here, '..', 'addon', 'ptr_napi.node');
        try {
            addon = require(file);
        } catch (e) {
            throw new Error(`cannot load ${file} (build it with \`make -C path-tracer_amd/js/addon\`; ` +
                            `there is no CPU fallback): ${e.message}`);
        }
    }
    return addon;
}
