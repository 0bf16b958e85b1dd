/*
 * webgl_stub.js — page script for oracle/make_golden_webgl.py.  TEST INFRASTRUCTURE, BUILD CONTAINER ONLY.
 *
 * It is loaded by the headless Chromium that ships inside the `kaleido` Python package (its `plotly` scope loads
 * whatever file --plotlyjs= names and then calls window.Plotly.toImage(figure) for every request line).  That
 * Chromium has a real WebGL 1 implementation (ANGLE on SwiftShader) with OES_texture_float,
 * WEBGL_color_buffer_float and EXT_float_blend, which is all the reference's particle pusher needs.  This file
 * pretends to be plotly.js only as far as the scope looks (a version string and toImage), and uses the request's
 * `layout.job` as a small command language:
 *
 *   {kind:'probe'}                         what the GL implementation says about itself
 *   {kind:'pic',  ...scene}                load the reference's UNMODIFIED utilities.js / matrix_webgl.js / spindle.js /
 *                                          empic.js from job.ref_dir at run time (synchronous XMLHttpRequest of
 *                                          file:// URLs + an AMD define() shim), run its own factory, set(),
 *                                          painters, precalc() and job.frames x (step(), density()), and keep every
 *                                          frame buffer after every stage, read back through the reference's own
 *                                          fb.readPixels (utilities.js:701-711)
 *   {kind:'sor',  ...case}                 the same for matrix_webgl.makeSORIterative
 *   {kind:'fetch', name, offset, count}    hand a piece of a kept snapshot to the driver (base64 of float32)
 *   {kind:'drop'}                          forget the snapshots
 *
 * Nothing of the reference is stored here or written anywhere by this file: it only ever returns numbers.
 * The randomness the reference draws from window.crypto.getRandomValues and Math.random (empic.js:148-173, quirk
 * Q8) is replaced by the LCG oracle/make_golden.js uses, so that a scene is reproducible and its entropy table can
 * be regenerated from a seed instead of being stored.
 */
(function () {
    'use strict';

    var FBO_NAMES = ['E', 'B', 'sink_mask', 'inv_cdf', 'B_loop_half', 'B_loop_tenth', 'R1', 'R2', 'R3', 'A',
        'position_A', 'velocity_A', 'position_B', 'velocity_B', 'rand_A', 'rand_B',
        'moments01', 'moments01_norm', 'moments01_avgA', 'moments01_avgB'];
    var TEX_NAMES = ['position_tex', 'velocity_tex', 'entropy_tex', 'rand_tex', 'E_tex', 'B_tex',
        'sink_mask_tex', 'inv_cdf_tex', 'shape_tex'];
    var SOR_FBO_NAMES = ['x_guess', 'x_result', 'x_stats', 'R', 'C', 'mv_product'];

    var snaps = {};          // name -> Float32Array
    var lcg = 12345;
    function nextU32() { lcg = (Math.imul(lcg, 1664525) + 1013904223) >>> 0; return lcg; }

    var realRandom = Math.random;
    var realGetRandomValues = window.crypto.getRandomValues;
    function seedRandomness(seed) {
        lcg = seed >>> 0;
        window.crypto.getRandomValues = function (a) { for (var i = 0; i < a.length; i++) a[i] = nextU32(); return a; };
        Math.random = function () { return nextU32() / 4294967296; };
    }
    function restoreRandomness() {
        Math.random = realRandom;
        window.crypto.getRandomValues = realGetRandomValues;
    }

    // ------------------------------------------------------------ AMD loader over file:// URLs
    function makeLoader(refDir) {
        var registry = {};
        function load(name) {
            if (registry[name]) return registry[name];
            var xhr = new XMLHttpRequest();
            xhr.open('GET', 'file://' + refDir + '/public/javascripts/' + name + '.js', false);
            xhr.send(null);
            var src = xhr.responseText;
            if (!src) throw new Error('could not read ' + name + '.js under ' + refDir);
            var captured = null;
            window.define = function (deps, fn) {
                if (typeof deps === 'function') { fn = deps; deps = []; }
                captured = { deps: deps, fn: fn };
            };
            (0, eval)(src);
            delete window.define;
            if (!captured) throw new Error('no define() in ' + name);
            var args = captured.deps.map(load);
            registry[name] = captured.fn.apply(null, args);
            return registry[name];
        }
        return load;
    }

    // util.webGL wrapped so that the frame buffers and texture arrays the factory creates can be found again
    // (creation order = the name lists above, as in oracle/make_golden.js)
    function instrument(util, fboNames, texNames) {
        var found = { fbos: [], texs: [], gl: null, wrappers: [] };
        var orig = util.webGL;
        util.webGL = function (canvas) {
            var w = orig(canvas);
            found.gl = w.gl;
            found.wrappers.push(w);
            var addTex = w.addTextureArray, addFbo = w.addFrameBuffer;
            var inFbo = false;
            w.addTextureArray = function (p) {
                var t = addTex(p);
                if (!inFbo) { t.__name = texNames ? texNames[found.texs.length] : undefined; found.texs.push(t); }
                return t;
            };
            w.addFrameBuffer = function (p) {
                inFbo = true;
                var f;
                try { f = addFbo(p); } finally { inFbo = false; }
                f.__name = fboNames[found.fbos.length];
                found.fbos.push(f);
                return f;
            };
            return w;
        };
        found.restore = function () { util.webGL = orig; };
        return found;
    }

    function readFbo(fb) {
        var a = new Float32Array(4 * fb.width * fb.height);
        fb.readPixels(a);
        return a;
    }

    function b64(f32, offset, count) {
        var bytes = new Uint8Array(f32.buffer, f32.byteOffset + 4 * offset, 4 * count);
        var parts = [], CH = 0x6000;
        for (var i = 0; i < bytes.length; i += CH) parts.push(String.fromCharCode.apply(null, bytes.subarray(i, Math.min(i + CH, bytes.length))));
        return btoa(parts.join(''));
    }
    function unb64(s) {
        var bin = atob(s), u8 = new Uint8Array(bin.length);
        for (var i = 0; i < bin.length; i++) u8[i] = bin.charCodeAt(i);
        return new Float32Array(u8.buffer);
    }
    // [n*k] float32 -> nested [n][k] of doubles (exactly the float32 values)
    function nested(f32, k) {
        var out = [], n = f32.length / k;
        for (var i = 0; i < n; i++) { var row = []; for (var c = 0; c < k; c++) row.push(f32[k * i + c]); out.push(row); }
        return out;
    }
    // [nr*nz*k] float32, index (i*nz + j)*k + c -> value[i][j] (k = 1) or value[i][j][c]
    function nestedGrid(f32, nr, nz, k) {
        var out = [];
        for (var i = 0; i < nr; i++) {
            var row = [];
            for (var j = 0; j < nz; j++) {
                if (k === 1) row.push(f32[i * nz + j]);
                else { var v = []; for (var c = 0; c < k; c++) v.push(f32[(i * nz + j) * k + c]); row.push(v); }
            }
            out.push(row);
        }
        return out;
    }

    function glInfo(gl) {
        var p = gl.getShaderPrecisionFormat(gl.FRAGMENT_SHADER, gl.HIGH_FLOAT);
        var dbg = gl.getExtension('WEBGL_debug_renderer_info');
        return {
            version: gl.getParameter(gl.VERSION), shading_language: gl.getParameter(gl.SHADING_LANGUAGE_VERSION),
            renderer: gl.getParameter(gl.RENDERER), vendor: gl.getParameter(gl.VENDOR),
            unmasked_renderer: dbg ? gl.getParameter(dbg.UNMASKED_RENDERER_WEBGL) : null,
            unmasked_vendor: dbg ? gl.getParameter(dbg.UNMASKED_VENDOR_WEBGL) : null,
            subpixel_bits: gl.getParameter(gl.SUBPIXEL_BITS),
            point_size_range: Array.prototype.slice.call(gl.getParameter(gl.ALIASED_POINT_SIZE_RANGE)),
            highp_fragment: [p.rangeMin, p.rangeMax, p.precision],
            OES_texture_float: !!gl.getExtension('OES_texture_float'),
            WEBGL_color_buffer_float: !!gl.getExtension('WEBGL_color_buffer_float'),
            EXT_float_blend: !!gl.getExtension('EXT_float_blend'),
            user_agent: navigator.userAgent,
        };
    }

    // ------------------------------------------------------------ jobs
    function jobProbe() {
        var c = document.createElement('canvas');
        var gl = c.getContext('webgl') || c.getContext('experimental-webgl');
        if (!gl) return { have_webgl: false };
        var info = glInfo(gl);
        info.have_webgl = true;
        return info;
    }

    function jobPic(job) {
        var load = makeLoader(job.ref_dir);
        var util = load('utilities');
        var found = instrument(util, FBO_NAMES, TEX_NAMES);
        var index = {}, errors = [];
        function keep(stage, names) {
            names.forEach(function (nm) {
                var fb = null;
                for (var i = 0; i < found.fbos.length; i++) if (found.fbos[i].__name === nm) fb = found.fbos[i];
                var a = readFbo(fb);
                snaps[stage + '/' + nm] = a;
                index[stage + '/' + nm] = a.length;
            });
            var e = found.gl.getError();
            if (e) errors.push([stage, e]);
        }
        seedRandomness(job.seed);
        try {
            var empic = load('empic');
            var sim = empic.makeCylindricalParticlePusher(job.spec);
            var tex = {};
            found.texs.forEach(function (t) { tex[t.__name] = t; });
            snaps['init/rand0'] = new Float32Array(tex.rand_tex.array);
            index['init/rand0'] = snaps['init/rand0'].length;
            snaps['init/entropy_head'] = new Float32Array(tex.entropy_tex.array.subarray(0, 64));
            index['init/entropy_head'] = 64;
            snaps['init/stamp'] = new Float32Array(tex.shape_tex.array);
            index['init/stamp'] = snaps['init/stamp'].length;
            var nr = job.spec.nr, nz = job.spec.nz;
            var value = {};
            if (job.position) value.position = nested(unb64(job.position), 3);
            if (job.velocity) value.velocity = nested(unb64(job.velocity), 3);
            if (job.E) value.E = nestedGrid(unb64(job.E), nr, nz, 3);
            if (job.B) value.B = nestedGrid(unb64(job.B), nr, nz, 3);
            if (job.sink_mask) value.sink_mask = nestedGrid(unb64(job.sink_mask), nr, nz, 1);
            if (job.source_pdf) value.source_pdf = nestedGrid(unb64(job.source_pdf), nr, nz, 1);
            // inputs given as JSON numbers (doubles) take precedence: the swgl_* scenes' inputs are not float32
            ['position', 'velocity', 'E', 'B', 'sink_mask', 'source_pdf'].forEach(function (k) {
                if (job[k + '_json']) value[k] = job[k + '_json'];
            });
            sim.set(value);
            snaps['set/inv_cdf_tex'] = new Float32Array(tex.inv_cdf_tex.array);
            index['set/inv_cdf_tex'] = snaps['set/inv_cdf_tex'].length;
            keep('set', ['position_A', 'velocity_A', 'rand_A', 'E', 'B', 'sink_mask', 'inv_cdf']);
            (job.painters || []).forEach(function (c) { sim[c[0]].apply(sim, c.slice(1)); });
            keep('painted', ['E', 'B']);
            sim.precalc();
            keep('precalc', ['R1', 'R2', 'R3', 'A']);
            for (var k = 1; k <= job.frames; k++) {
                sim.step();
                keep('step' + k, ['position_A', 'velocity_A', 'rand_A']);
                sim.density();
                keep('density' + k, ['moments01', 'moments01_norm', 'moments01_avgA', 'moments01_avgB']);
            }
            return { index: index, gl_errors: errors, n_fbos: found.fbos.length, n_texs: found.texs.length,
                api: Object.keys(sim).sort(), gl: glInfo(found.gl) };
        } finally {
            restoreRandomness();
            found.restore();
        }
    }

    function jobSor(job) {
        var load = makeLoader(job.ref_dir);
        var util = load('utilities');
        var found = instrument(util, SOR_FBO_NAMES, null);
        var quietLog = console.log;
        console.log = function () {};            // solve() prints R, C and every iterate
        try {
            var mw = load('matrix_webgl');
            var canvas = document.createElement('canvas');
            canvas.width = canvas.height = Math.pow(2, job.n_power);
            var webgl = util.webGL(canvas);
            var spec = { n_power: job.n_power, webgl: webgl };
            if (job.relaxation !== null && job.relaxation !== undefined) spec.relaxation = job.relaxation;
            var eq = mw.makeSORIterative(spec);
            var L = eq.vec_length;
            var Aflat = unb64(job.A), b = Array.prototype.slice.call(unb64(job.b)), x0 = Array.prototype.slice.call(unb64(job.x0));
            var A = [];
            for (var r = 0; r < L; r++) { A.push([]); for (var c = 0; c < L; c++) A[r].push(Aflat[c + L * r]); }
            var by = {};
            found.fbos.forEach(function (f) { by[f.__name] = f; });
            var out = { vec_length: L, vec_height: eq.vec_height, calls: [], gl_errors: [] };
            var pre = job.name + '/';
            eq.set_matrix(A).set_b(b).init_vector(x0);
            snaps[pre + 'x_after_init'] = readFbo(by.x_result);
            job.calls.forEach(function (params, ci) {
                var res = eq.solve(params);
                var tag = pre + 'call' + ci + '/';
                snaps[tag + 'result'] = new Float32Array(res.result);
                ['x_result', 'x_guess', 'x_stats', 'R', 'C'].forEach(function (nm) { snaps[tag + nm] = readFbo(by[nm]); });
                out.calls.push({ params: params, correlation: (res.correlation !== res.correlation) ? 'NaN' : res.correlation,
                    diff: res.diff, iterations: res.iterations });
                var e = found.gl.getError();
                if (e) out.gl_errors.push([ci, e]);
            });
            out.index = {};
            Object.keys(snaps).forEach(function (k) { if (k.indexOf(pre) === 0) out.index[k] = snaps[k].length; });
            out.gl = glInfo(found.gl);
            return out;
        } finally {
            console.log = quietLog;
            found.restore();
        }
    }

    function run(job) {
        if (job.kind === 'probe') return jobProbe();
        if (job.kind === 'pic') return jobPic(job);
        if (job.kind === 'sor') return jobSor(job);
        if (job.kind === 'fetch') {
            var a = snaps[job.name];
            if (!a) throw new Error('no snapshot ' + job.name);
            var count = Math.min(job.count, a.length - job.offset);
            return { name: job.name, offset: job.offset, count: count, data: b64(a, job.offset, count) };
        }
        if (job.kind === 'drop') { snaps = {}; return { dropped: true }; }
        throw new Error('unknown job kind ' + job.kind);
    }

    window.Plotly = {
        version: '2.0.0',
        toImage: function (figure) {
            var job = (figure && figure.layout && figure.layout.job) || { kind: 'probe' };
            try {
                return Promise.resolve({ ok: true, reply: run(job) });
            } catch (e) {
                return Promise.resolve({ ok: false, error: String(e), stack: e && e.stack ? String(e.stack).slice(0, 2000) : null });
            }
        },
    };
})();
