/*
 * swgl.js — a software stand-in for the object the reference's util.webGL(canvas) returns.
 *
 * TEST INFRASTRUCTURE ONLY (see glsl_eval.js for what the results do and do not pin).
 * It implements just what the reference's particle-pusher factory uses
 * (utilities.js:133-760): float RGBA textures with NEAREST/CLAMP lookups, frame buffers,
 * vertex buffers, programs with set()/draw(), full-target triangle pairs, point sprites and
 * ONE,ONE blending.  Shaders are the strings the reference passes to linkProgram, evaluated
 * by glsl_eval.js.
 *
 * Rasterisation conventions (OpenGL ES 2.0 sections 3.3, 3.5; exact where the spec is exact):
 *   - the two triangles of a draw must cover the whole target; varyings are the affine
 *     interpolation of the vertex outputs at each pixel centre, rounded to float32;
 *   - a point sprite covers the pixel centres inside the square of side gl_PointSize around
 *     its window position; gl_PointCoord = (1/2 + (xf + 1/2 - xw)/size,
 *     1/2 - (yf + 1/2 - yw)/size); points whose centre is outside the clip volume are dropped;
 *   - a pixel centre exactly ON the square's edge (window position a multiple of half a pixel:
 *     in this path only particles exactly at r = 0) is implementation-defined in GL ES.  It is
 *     resolved here as the limit of the point displaced by +0 in x and y: footprint
 *     floor(xw)-5 .. floor(xw)+5 and stamp texel di+5, which is also what the restatement
 *     does.  A top-left fill rule would give floor(xw)-6 .. floor(xw)+4 (DESIGN.md Q13);
 *   - blending and clears act in float32; draws without a target (the canvas) are skipped.
 */
'use strict';
const glsl = require('./glsl_eval.js');
const f = Math.fround;

function makeSoftwareGL(names) {
    names = names || {};
    const state = { fbos: [], texs: [], progs: [], draws: 0 };
    const gl = {};

    gl.enableFloatTexture = function () {};

    gl.addVertexData = function (array) {
        const ncomp = Array.isArray(array[0]) ? array[0].length : 1;
        const data = new Float32Array(ncomp * array.length);
        for (let i = 0; i < array.length; i++) {
            if (ncomp === 1) data[i] = array[i]; else for (let c = 0; c < ncomp; c++) data[ncomp * i + c] = array[i][c];
        }
        const buff = { kind: 'vertex', ncomp: ncomp, data: data, count: array.length };
        buff.bind = function (prog, name) { prog.attribs[name] = buff; return buff; };
        return buff;
    };

    function makeTexture(params, list, nameList) {
        const n = 4 * params.width * params.height;
        const tex = { width: params.width, height: params.height, source: params.array,
            array: params.array ? new Float32Array(params.array) : new Float32Array(n) };
        tex.name = nameList ? nameList[list.length] : undefined;
        tex.bind = function (prog, name) { prog.samplers[name] = tex; return tex; };
        tex.update = function () { tex.array.set(tex.source); };
        list.push(tex);
        return tex;
    }
    gl.addTextureArray = function (params) { const t = makeTexture(params, state.texs, names.tex); t.kind = 'tex'; return t; };
    gl.addFrameBuffer = function (params) {
        const t = makeTexture(params, state.fbos, names.fbo);
        t.kind = 'fbo';
        t.readPixels = function (array) { array.set(t.array); };
        t.clear = function (r, g, b, a) { for (let k = 0; k < t.array.length; k += 4) { t.array[k] = r; t.array[k + 1] = g; t.array[k + 2] = b; t.array[k + 3] = a; } };
        return t;
    };
    gl.clear = function () {};
    gl.bindCanvas = function () {};

    gl.linkProgram = function (spec) {
        const prog = { vs: glsl.parse(spec.vertexShaderSource), fs: glsl.parse(spec.fragmentShaderSource),
            uniforms: {}, samplers: {}, attribs: {} };
        prog.name = names.prog ? names.prog[state.progs.length] : undefined;
        state.progs.push(prog);

        prog.set = function (obj) {
            for (const k in obj) {
                const v = obj[k];
                const known = (k in prog.vs.decls.uniform) || (k in prog.fs.decls.uniform) || (k in prog.vs.decls.attribute);
                if (!known) throw new Error('Could not find uniform: ' + k);
                if (typeof v === 'number') prog.uniforms[k] = f(v);
                else if (Array.isArray(v)) prog.uniforms[k] = v.map(f);
                else if (v && v.bind) v.bind(prog, k);
                else throw new Error('Cannot add uniform value: ' + k);
            }
            return prog;
        };

        function baseEnv() { return Object.assign({}, prog.uniforms, prog.samplers); }
        function runVertex(index) {
            const env = baseEnv();
            for (const a in prog.attribs) {
                const b = prog.attribs[a];
                env[a] = b.ncomp === 1 ? b.data[index] : Array.prototype.slice.call(b.data, b.ncomp * index, b.ncomp * (index + 1));
            }
            glsl.run(prog.vs, env);
            return env;
        }
        function runFragment(varyings, pointCoord) {
            const env = Object.assign(baseEnv(), varyings);
            if (pointCoord) env.gl_PointCoord = pointCoord;
            glsl.run(prog.fs, env);
            return env.gl_FragColor;
        }
        function write(out, o, color, blend) {
            for (let c = 0; c < 4; c++) out[o + c] = blend ? f(color[c] + out[o + c]) : color[c];
        }

        prog.draw = function (params) {
            state.draws++;
            const target = params.target;
            if (!target) return prog;                       // the canvas: presentation only
            const W = target.width, H = target.height;
            if (params.blend && (params.blend[0] !== 'ONE' || params.blend[1] !== 'ONE')) throw new Error('swgl: only ONE,ONE blending');
            if (params.clear_color) target.clear(f(params.clear_color[0]), f(params.clear_color[1]), f(params.clear_color[2]), f(params.clear_color[3]));
            // a draw may sample its own target only through a different texture object; render
            // into a copy so every fragment sees the pre-draw contents
            const out = new Float32Array(target.array);
            const varyingNames = Object.keys(prog.vs.decls.varying);

            if (params.triangles) {
                if (params.triangles !== 6) throw new Error('swgl: expected one quad');
                const v = [0, 1, 2, 3, 4, 5].map(runVertex);
                const corner = {};
                v.forEach(function (e) { corner[e.gl_Position[0] + ',' + e.gl_Position[1]] = e; });
                const bl = corner['-1,-1'], br = corner['1,-1'], tl = corner['-1,1'], tr = corner['1,1'];
                if (!bl || !br || !tl || !tr) throw new Error('swgl: triangles must cover the target');
                for (let j = 0; j < H; j++) {
                    const ty = (j + 0.5) / H;
                    for (let i = 0; i < W; i++) {
                        const tx = (i + 0.5) / W;
                        const varyings = {};
                        varyingNames.forEach(function (name) {
                            const a = bl[name], bx = br[name], by = tl[name];
                            const lerp = function (c) { return f(a[c] + (bx[c] - a[c]) * tx + (by[c] - a[c]) * ty); };
                            varyings[name] = Array.isArray(a) ? a.map(function (_, c) { return lerp(c); }) : f(a + (bx - a) * tx + (by - a) * ty);
                        });
                        write(out, 4 * (i + W * j), runFragment(varyings, null), params.blend);
                    }
                }
            }

            if (params.points) {
                for (let k = 0; k < params.points; k++) {
                    const e = runVertex(k);
                    const p = e.gl_Position, size = e.gl_PointSize;
                    const w = p[3];
                    if (!(p[0] >= -w && p[0] <= w && p[1] >= -w && p[1] <= w && p[2] >= -w && p[2] <= w)) continue;
                    // viewport transform: exact here; in float32, step by step, under the 'gpu' convention (where a
                    // point within an ulp of a pixel edge can land in the neighbouring cell)
                    const gpu = glsl.convention() === 'gpu';
                    const xw = gpu ? f(f(f(p[0] / w) + 1) * f(0.5 * W)) : (p[0] / w + 1) * 0.5 * W;
                    const yw = gpu ? f(f(f(p[1] / w) + 1) * f(0.5 * H)) : (p[1] / w + 1) * 0.5 * H;
                    const varyings = {};
                    varyingNames.forEach(function (name) { varyings[name] = e[name]; });
                    // pixel centres with xw - size/2 < c <= xw + size/2 (ties: see header)
                    const x0 = Math.max(0, Math.floor(xw - size / 2 - 0.5) + 1), x1 = Math.min(W - 1, Math.floor(xw + size / 2 - 0.5));
                    const y0 = Math.max(0, Math.floor(yw - size / 2 - 0.5) + 1), y1 = Math.min(H - 1, Math.floor(yw + size / 2 - 0.5));
                    const untie = function (s, dir) {
                        const q = s * size, n = Math.round(q);
                        return Math.abs(q - n) < 1e-9 ? (n + 0.5 * dir) / size : s;
                    };
                    for (let py = y0; py <= y1; py++) {
                        for (let px = x0; px <= x1; px++) {
                            const pc = [f(untie(0.5 + (px + 0.5 - xw) / size, -1)), f(untie(0.5 - (py + 0.5 - yw) / size, +1))];
                            write(out, 4 * (px + W * py), runFragment(varyings, pc), params.blend);
                        }
                    }
                }
            }
            target.array.set(out);
            return prog;
        };
        return prog;
    };

    return { gl: gl, state: state };
}

module.exports = { makeSoftwareGL: makeSoftwareGL };
