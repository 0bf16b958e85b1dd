#!/usr/bin/env node
/*
 * make_golden.js — generates tests/golden/* by IMPORTING the reference's own host
 * JavaScript (read-only, from /root/reference) under Node.  TEST INFRASTRUCTURE.
 *
 *   node oracle/make_golden.js [/root/reference] [tests/golden]
 *
 * What it does: evaluates utilities.js, spindle.js and empic.js through an AMD
 * define() shim, replaces util.webGL with a recorder (Node has no WebGL: no shader is
 * executed HERE — oracle/make_golden_webgl.py runs the reference under a real WebGL in the
 * headless Chromium of the kaleido package and writes the webgl_* fixtures), calls the
 * reference factory and its
 * set()/step()/density() methods, and writes ONLY numeric data and call-order
 * metadata: Float32Array contents the reference computed on the host, uniform
 * values it set, and which resource each draw reads and writes.  No reference
 * source text (JS or GLSL) is written anywhere.
 *
 * The swgl_* fixtures (section 8) are different in kind: there util.webGL is replaced by
 * oracle/swgl.js, which EVALUATES the shader strings the reference hands to linkProgram
 * (oracle/glsl_eval.js), so the reference's own set/precalc/step/density code and shader text
 * run end to end in software.  They check the restatement's transcription of the shaders;
 * see glsl_eval.js for what they cannot pin.
 *
 * The fixtures travel to the GPU box; /root/reference does not.
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const zlib = require('zlib');
const makeSoftwareGL = require('./swgl.js').makeSoftwareGL;

const refRoot = process.argv[2] || '/root/reference';
const outDir = process.argv[3] || path.join(__dirname, '..', 'tests', 'golden');
const jsDir = path.join(refRoot, 'public', 'javascripts');
fs.mkdirSync(outDir, { recursive: true });

// ---------------------------------------------------------------- AMD loader
function loadAmd(name, registry, sandboxExtra) {
    if (registry[name]) return registry[name];
    const src = fs.readFileSync(path.join(jsDir, name + '.js'), 'utf8');
    let captured = null;
    const sandbox = Object.assign({
        define: function (deps, fn) {
            if (typeof deps === 'function') { fn = deps; deps = []; }
            captured = { deps: deps, fn: fn };
        },
        console: console, Math: Math, Float32Array: Float32Array, Uint32Array: Uint32Array,
        Uint8Array: Uint8Array, Array: Array, Error: Error, Object: Object, Date: Date,
    }, sandboxExtra);
    sandbox.window = sandbox.window || sandbox;
    vm.runInNewContext(src, sandbox, { filename: name + '.js' });
    if (!captured) throw new Error('no define() in ' + name);
    const args = captured.deps.map(function (d) { return loadAmd(d, registry, sandboxExtra); });
    registry[name] = captured.fn.apply(null, args);
    return registry[name];
}

// ---------------------------------------------------------------- recorder
// Names follow creation order in empic.js (frame buffers :186-291, :499-502,
// :666-672, :933, :1040, :1071-1072; texture arrays :125-241, :973; programs
// :295-1141).
const FBO_NAMES = ['E', 'B', 'sink_mask', 'inv_cdf', 'B_loop_half', 'B_loop_tenth', 'R1', 'R2', 'R3', 'A',
    'position_A', 'velocity_A', 'position_B', 'velocity_B', 'rand_A', 'rand_B',
    'moments01', 'moments01_norm', 'moments01_avgA', 'moments01_avgB'];
const TEX_NAMES = ['position_tex', 'velocity_tex', 'entropy_tex', 'rand_tex', 'E_tex', 'B_tex',
    'sink_mask_tex', 'inv_cdf_tex', 'shape_tex'];
const PROG_NAMES = ['CurrentLoopShape', 'CurrentLoop', 'CurrentZ', 'BZ', 'BTheta', 'BMag',
    'Pre1', 'Pre2', 'Pre3', 'PreA', 'StepRandB', 'StepVelocityB', 'StepPositionB',
    'StepRandA', 'StepVelocityA', 'StepPositionA', 'Moments01', 'NormalizeMoments01',
    'AvgMoments', 'Density', 'Set'];

function makeRecorder() {
    const rec = { draws: [], fbos: [], texs: [], progs: [], literals: {} };
    const webgl = {};
    webgl.enableFloatTexture = function () {};
    webgl.addVertexData = function (array) { return { kind: 'vertex', length: array.length, bind: function () {} }; };
    webgl.addTextureArray = function (params) {
        const t = { kind: 'tex', name: TEX_NAMES[rec.texs.length], width: params.width, height: params.height,
            array: params.array, updates: 0 };
        t.update = function () { t.updates++; };
        rec.texs.push(t);
        return t;
    };
    webgl.addFrameBuffer = function (params) {
        const f = { kind: 'fbo', name: FBO_NAMES[rec.fbos.length], width: params.width, height: params.height };
        rec.fbos.push(f);
        return f;
    };
    webgl.linkProgram = function (params) {
        const prog = { name: PROG_NAMES[rec.progs.length], uniforms: {}, samplers: {} };
        // numeric literals the factory baked into this program's text: keep the
        // numbers only (N(x) = toFixed(20) prints exactly 20 decimals)
        const lits = (params.fragmentShaderSource.match(/-?\d+\.\d{20}/g) || []).map(Number);
        if (lits.length) rec.literals[prog.name] = lits;
        prog.set = function (obj) {
            for (const k in obj) {
                const v = obj[k];
                if (typeof v === 'number') prog.uniforms[k] = v;
                else if (v && (v.kind === 'tex' || v.kind === 'fbo')) prog.samplers[k] = v.name;
            }
            return prog;
        };
        prog.draw = function (p) {
            rec.draws.push({
                program: prog.name,
                target: p.target ? p.target.name : 'canvas',
                reads: Object.assign({}, prog.samplers),
                uniforms: Object.assign({}, prog.uniforms),
                blend: p.blend || null,
                clear_color: p.clear_color || null,
                triangles: p.triangles || 0,
                points: p.points || 0,
            });
            return prog;
        };
        rec.progs.push(prog);
        return prog;
    };
    return { rec: rec, webgl: webgl };
}

// deterministic stand-ins for window.crypto / Math.random (quirk Q8): values are
// irrelevant to every fixture written below.
let lcg = 12345;
function nextU32() { lcg = (Math.imul(lcg, 1664525) + 1013904223) >>> 0; return lcg; }

function makeReference(software) {
    const r = software ? null : makeRecorder();
    const sw = software ? makeSoftwareGL({ fbo: FBO_NAMES, tex: TEX_NAMES, prog: PROG_NAMES }) : null;
    const extra = {
        document: { createElement: function () { return { style: {} }; }, body: { appendChild: function () {} } },
        crypto: { getRandomValues: function (a) { for (let i = 0; i < a.length; i++) a[i] = nextU32(); } },
    };
    if (software) {
        // Math.random seeds the per-particle random state (empic.js:168-173)
        const m = Object.create(Math);
        m.random = function () { return nextU32() / 4294967296; };
        extra.Math = m;
    }
    const registry = {};
    const util = loadAmd('utilities', registry, extra);
    util.webGL = function () { return software ? sw.gl : r.webgl; };
    const empic = loadAmd('empic', registry, extra);
    return { empic: empic, util: util, rec: r && r.rec, sw: sw && sw.state };
}

function f32list(a) { return Array.prototype.slice.call(a); }
function writeJson(name, obj) { fs.writeFileSync(path.join(outDir, name), JSON.stringify(obj, null, 1) + '\n'); }
function writeF32gz(name, arr) {
    const buf = Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength);
    fs.writeFileSync(path.join(outDir, name), zlib.gzipSync(buf, { level: 9 }));
}

// ---------------------------------------------------------------- fixtures
const specs = {
    demo: { radius: 1, height: 2, nr: 400, nz: 800, dt: 2e-9, nparticles: 4, particle_mass: 1.67e-27, particle_charge: 1.602e-19 },
    squat: { radius: 0.35, height: 0.2, nr: 24, nz: 16, dt: 5e-10, nparticles: 3, particle_mass: 9.109e-31, particle_charge: -1.602e-19 },
    c1: { radius: 1, height: 1, nr: 128, nz: 128, dt: 2e-9, nparticles: 4, particle_mass: 1.67e-27, particle_charge: 1.602e-19 },
};

// (1) constants, uniforms and shader literals per spec; (2) stamp
const constants = {};
let stamp = null;
for (const key in specs) {
    const ref = makeReference();
    const spec = specs[key];
    ref.empic.makeCylindricalParticlePusher(spec);
    const byName = {};
    ref.rec.progs.forEach(function (p) { byName[p.name] = p; });
    constants[key] = {
        spec: spec,
        u_h: byName.Pre1.uniforms.u_h,
        u_step_factor: byName.StepPositionB.uniforms.u_step_factor,
        u_pointsize: byName.Moments01.uniforms.u_pointsize,
        u_ratio: byName.AvgMoments.uniforms.u_ratio,
        literal_frz_Pre1: ref.rec.literals.Pre1,
        literal_frz_Pre2: ref.rec.literals.Pre2,
        literal_fzr_Pre3: ref.rec.literals.Pre3,
        literal_fr_fr_fz_PreA: ref.rec.literals.PreA,
        n_programs: ref.rec.progs.length,
        particle_count: ref.rec.texs[0].array.length / 4,
    };
    if (!stamp) stamp = f32list(ref.rec.texs[8].array).filter(function (_, i) { return i % 4 === 0; });
}
writeJson('constants.json', constants);
writeJson('stamp.json', { nshape: 11, red: stamp });

// (3) particle upload, (4) grid packing, (6) draw order: one small instance
(function () {
    const spec = specs.squat;
    const ref = makeReference();
    const sim = ref.empic.makeCylindricalParticlePusher(spec);
    const n = spec.nparticles * spec.nparticles;
    const pos = [], vel = [], E = [], B = [], sink = [], pdf = [];
    let s = 7;
    function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
    for (let p = 0; p < n; p++) {
        pos.push([0.3 * (rnd() - 0.5), 0.3 * (rnd() - 0.5), 0.2 * rnd()]);
        vel.push([0.01 * (rnd() - 0.5), 0.01 * (rnd() - 0.5), 0.01 * (rnd() - 0.5)]);
    }
    pos[0] = [0.1, 0.2, 0.1]; // 0.1*(1/0.35) is not exact in float
    for (let i = 0; i < spec.nr; i++) {
        E.push([]); B.push([]); sink.push([]); pdf.push([]);
        for (let j = 0; j < spec.nz; j++) {
            E[i].push([1e3 * rnd(), -2e3 * rnd(), 5e2 * (rnd() - 0.5)]);
            B[i].push([0.1 * (rnd() - 0.5), 0.2 * (rnd() - 0.5), 1.0 * rnd()]);
            sink[i].push((i === spec.nr - 1 || j === 0 || j === spec.nz - 1) ? 0 : 1);
            pdf[i].push(rnd());
        }
    }
    const drawsBefore = ref.rec.draws.length;
    sim.set({ E: E, B: B, position: pos, velocity: vel, sink_mask: sink, source_pdf: pdf });
    const setDraws = ref.rec.draws.slice(drawsBefore);
    const tex = {};
    ref.rec.texs.forEach(function (t) { tex[t.name] = t; });
    writeJson('upload_squat.json', {
        spec: spec,
        position_in: pos, velocity_in: vel,
        position_arr: f32list(tex.position_tex.array),
        velocity_arr: f32list(tex.velocity_tex.array),
        E_in: E, B_in: B, sink_in: sink,
        E_arr: f32list(tex.E_tex.array), B_arr: f32list(tex.B_tex.array),
        sink_mask_arr: f32list(tex.sink_mask_tex.array),
        set_draws: setDraws.map(function (d) { return { program: d.program, target: d.target, reads: d.reads }; }),
    });
    // the random pdf on 24x16 is strictly positive -> full table, no NaN
    writeF32gz('inv_cdf_squat_random.f32.gz', tex.inv_cdf_tex.array);
    writeJson('inv_cdf_squat_random.json', { nr: spec.nr, nz: spec.nz, pdf: pdf, layout: '4*(i + 512*j) + c, c=0:x c=1:y', file: 'inv_cdf_squat_random.f32.gz' });

    // painters + precalc + step + density: order and bindings
    let mark = ref.rec.draws.length;
    sim.addCurrentLoop(0.3, 0.1, 1e6);
    sim.addCurrentZ(2e5);
    sim.addBZ(0.25);
    sim.addBTheta(-0.125);
    const painterDraws = ref.rec.draws.slice(mark);
    mark = ref.rec.draws.length;
    sim.precalc();
    const precalcDraws = ref.rec.draws.slice(mark);
    mark = ref.rec.draws.length;
    sim.step();
    const stepDraws = ref.rec.draws.slice(mark);
    mark = ref.rec.draws.length;
    sim.density();
    const densityDraws = ref.rec.draws.slice(mark);
    writeJson('draw_order.json', {
        api: Object.keys(sim).sort(),
        painters: painterDraws, precalc: precalcDraws, step: stepDraws, density: densityDraws,
    });
})();

// (5) inverse-CDF tables with the demo-shaped block source and with empty rows (quirk Q3)
(function () {
    const cases = {
        // scaled-down fusionsim.js:114-122 source block: rows < 5, columns 35..44 of 40x80
        block: { nr: 40, nz: 80, f: function (i, j) { return (i < 5 && j >= 35 && j < 45) ? 1.0 : 0.0; } },
        // ragged: empty rows inside and at the end, empty first column, uneven weights
        ragged: { nr: 16, nz: 12, f: function (i, j) {
            if (i === 3 || i === 4 || i >= 13) return 0.0;
            if (j === 0) return 0.0;
            return ((i * 7 + j * 3) % 5 === 0) ? 0.0 : 0.25 + ((i * 31 + j * 17) % 11) / 7.0;
        } },
        // uniform interior as SURVEY 8(d): strictly positive except the sink frame
        interior: { nr: 32, nz: 32, f: function (i, j) { return (i === 31 || j === 0 || j === 31) ? 0.0 : 1.0; } },
    };
    for (const key in cases) {
        const c = cases[key];
        const spec = { radius: 1, height: 1, nr: c.nr, nz: c.nz, dt: 2e-9, nparticles: 2, particle_mass: 1.67e-27, particle_charge: 1.602e-19 };
        const ref = makeReference();
        const sim = ref.empic.makeCylindricalParticlePusher(spec);
        const pdf = [];
        for (let i = 0; i < c.nr; i++) { pdf.push([]); for (let j = 0; j < c.nz; j++) pdf[i].push(c.f(i, j)); }
        sim.set({ source_pdf: pdf });
        const arr = ref.rec.texs[7].array;
        let nan = 0;
        for (let k = 0; k < arr.length; k++) if (arr[k] !== arr[k]) nan++;
        writeF32gz('inv_cdf_' + key + '.f32.gz', arr);
        writeJson('inv_cdf_' + key + '.json', { nr: c.nr, nz: c.nz, pdf: pdf, nan_count: nan,
            layout: '4*(i + 512*j) + c, c=0:x c=1:y', file: 'inv_cdf_' + key + '.f32.gz' });
    }
    // a pdf whose first row is empty makes the reference throw inside set()
    const spec = { radius: 1, height: 1, nr: 4, nz: 4, dt: 2e-9, nparticles: 2, particle_mass: 1.67e-27, particle_charge: 1.602e-19 };
    const ref = makeReference();
    const sim = ref.empic.makeCylindricalParticlePusher(spec);
    let threw = null;
    try { sim.set({ source_pdf: [[0, 0, 0, 0], [1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]] }); } catch (e) { threw = e.constructor.name; }
    writeJson('inv_cdf_throws.json', { pdf: [[0, 0, 0, 0], [1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]], threw: threw });
})();

// (7) spec validation messages (utilities.js:118-127)
(function () {
    const ref = makeReference();
    const msgs = {};
    const bad = {
        missing_radius: { height: 2, nr: 4, nz: 4, dt: 1e-9, nparticles: 2, particle_mass: 1, particle_charge: 1 },
        string_nr: { radius: 1, height: 2, nr: '4', nz: 4, dt: 1e-9, nparticles: 2, particle_mass: 1, particle_charge: 1 },
        missing_charge: { radius: 1, height: 2, nr: 4, nz: 4, dt: 1e-9, nparticles: 2, particle_mass: 1 },
    };
    for (const k in bad) {
        try { ref.empic.makeCylindricalParticlePusher(bad[k]); msgs[k] = null; } catch (e) { msgs[k] = e.message; }
    }
    writeJson('validation.json', msgs);
})();

// (8) the reference's shaders evaluated in software: small scenes, every pass
function swglScene(name, cfg, keepOnly) {
    lcg = cfg.seed;
    const spec = cfg.spec;
    const ref = makeReference(true);
    const sim = ref.empic.makeCylindricalParticlePusher(spec);
    const n = spec.nparticles * spec.nparticles;
    const by = {};
    ref.sw.fbos.concat(ref.sw.texs).forEach(function (t) { by[t.name] = t; });
    let s = cfg.input_seed;
    function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
    const pos = [], vel = [], E = [], B = [], sink = [], pdf = [];
    for (let p = 0; p < n; p++) {
        const rr = cfg.r_max * spec.radius * Math.sqrt(rnd()), th = 6.283185307179586 * rnd();
        pos.push([rr * Math.cos(th), rr * Math.sin(th), spec.height * (0.05 + 0.9 * rnd())]);
        vel.push([cfg.v * (rnd() - 0.5), cfg.v * (rnd() - 0.5), cfg.v * (rnd() - 0.5)]);
    }
    for (let i = 0; i < spec.nr; i++) {
        E.push([]); B.push([]); sink.push([]); pdf.push([]);
        for (let j = 0; j < spec.nz; j++) {
            E[i].push([cfg.E * (rnd() - 0.5), cfg.E * (rnd() - 0.5), cfg.E * (rnd() - 0.5)]);
            B[i].push([cfg.B * (rnd() - 0.5), cfg.B * (rnd() - 0.5), cfg.B * rnd()]);
            sink[i].push(cfg.sink(i, j));
            pdf[i].push(cfg.pdf(i, j, rnd));
        }
    }
    const chunks = [], index = {};
    let offset = 0;
    function snap(stage, names) {
        names.forEach(function (nm) {
            const a = new Float32Array(by[nm].array);
            index[stage + '/' + nm] = [offset, a.length];
            chunks.push(Buffer.from(a.buffer));
            offset += a.length;
        });
    }
    const rand0 = f32list(by.rand_tex.array);
    sim.set({ E: E, B: B, position: pos, velocity: vel, sink_mask: sink, source_pdf: pdf });
    snap('set', ['position_A', 'velocity_A', 'rand_A', 'E', 'B', 'sink_mask']);
    cfg.painters.forEach(function (c) { sim[c[0]].apply(sim, c.slice(1)); });
    snap('painted', ['E', 'B']);
    sim.precalc();
    snap('precalc', ['R1', 'R2', 'R3', 'A']);
    for (let k = 1; k <= cfg.frames; k++) {
        sim.step();
        snap('step' + k, ['position_A', 'velocity_A', 'rand_A']);
        sim.density();
        snap('density' + k, ['moments01', 'moments01_norm', 'moments01_avgA', 'moments01_avgB']);
    }
    if (keepOnly) { // convention study (section 10): hand the snapshots back, write nothing
        const all = new Float32Array(Buffer.concat(chunks).buffer.slice(0));
        const out = {};
        Object.keys(index).forEach(function (k) { out[k] = all.subarray(index[k][0], index[k][0] + index[k][1]); });
        return out;
    }
    fs.writeFileSync(path.join(outDir, name + '.f32.gz'), zlib.gzipSync(Buffer.concat(chunks), { level: 9 }));
    writeJson(name + '.json', {
        what: "outputs of the reference's own host code and shader strings evaluated by oracle/swgl.js + glsl_eval.js (float32 per operation); pins transcription, not GPU arithmetic",
        spec: spec, frames: cfg.frames, entropy_lcg_seed: cfg.seed,
        entropy_rule: 'u32 stream x <- 1664525*x + 1013904223 (mod 2^32), texel value = float32(x / 0xFFFFFFFF), 4*1024*1024 values in order',
        position_in: pos, velocity_in: vel, E_in: E, B_in: B, sink_in: sink, pdf_in: pdf, rand0: rand0,
        painters: cfg.painters,
        layout: 'RGBA float32, texel 4*(i + width*j)', file: name + '.f32.gz', index: index,
    });
}

// electrons in a squat cylinder, sink frame, block source: deaths and re-injection every frame
const SCENES = {};
SCENES.swgl_scene = {
    seed: 0x5EED0008, input_seed: 99, frames: 6, r_max: 0.94, v: 0.3, E: 2e5, B: 0.02,
    spec: { radius: 0.35, height: 0.2, nr: 24, nz: 16, dt: 5e-10, nparticles: 12, particle_mass: 9.109e-31, particle_charge: -1.602e-19 },
    sink: function (i, j) { return (i === 23 || j === 0 || j === 15) ? 0 : 1; },
    pdf: function (i, j, rnd) { return (i < 6 && j >= 5 && j < 11) ? 0.5 + rnd() : 0.0; },
    painters: [['addCurrentLoop', 0.2, 0.1, 4e4], ['addCurrentZ', 3e3], ['addBZ', 0.01], ['addBTheta', -0.005]],
};
swglScene('swgl_scene', SCENES.swgl_scene);
// protons in the demo's 1 x 2 m proportions (factor_r != factor_z), no sink on the outer walls so
// particles leave the unit square (clamped lookups, whole-point clipping in the deposit), an
// absorbing slab inside, and a source with empty rows (NaN sites of the inverse CDF, quirk Q3)
SCENES.swgl_tall = {
    seed: 0x5EED0009, input_seed: 4242, frames: 5, r_max: 0.99, v: 0.25, E: 3e6, B: 1.5,
    spec: { radius: 1, height: 2, nr: 20, nz: 40, dt: 2e-9, nparticles: 10, particle_mass: 1.67e-27, particle_charge: 1.602e-19 },
    sink: function (i, j) { return (i >= 8 && i < 12 && j >= 10 && j < 30) ? 0 : 1; },
    pdf: function (i, j, rnd) { return (i % 3 === 1 || j < 4) ? 0.0 : 0.25 + rnd(); },
    painters: [['addCurrentLoop', 0.8, 2.0, -1e7], ['addCurrentLoop', 0.8, 0.0, 1e7], ['addBZ', 0.3]],
};
swglScene('swgl_tall', SCENES.swgl_tall);

// (10) how far the results move under the OTHER plausible arithmetic conventions of a GPU's GLSL compiler
// (glsl_eval.js setConvention('gpu'): contracted multiply-adds, dot as an fma chain, division through a rounded
// reciprocal, float32 viewport transform).  Both runs execute the reference's own shader text; the numbers are the
// honest error bar on "matches the reference within 1e-3" for a real browser GPU, which no test here can reach.
(function conventionStudy() {
    const glslEval = require('./glsl_eval.js');
    const report = { what: "the reference's own host code and shader strings evaluated under two arithmetic conventions (oracle/glsl_eval.js): " +
        "'ieee' = one float32 rounding per operation, no contraction (the convention of the restatement and of the HIP kernels); 'gpu' = a*b+-c " +
        "contracted to fused multiply-adds, dot() as an fma chain, x/y as x*(1/y) with a rounded reciprocal, float32 viewport transform of point " +
        "sprites.  Per stage: largest difference relative to the largest magnitude of the stage's texture, NGP cells and alive flags that differ.",
        scenes: {} };
    Object.keys(SCENES).forEach(function (name) {
        const cfg = SCENES[name], spec = cfg.spec, n = spec.nparticles * spec.nparticles;
        glslEval.setConvention('ieee');
        const a = swglScene(name, cfg, true);
        glslEval.setConvention('gpu');
        const b = swglScene(name, cfg, true);
        glslEval.setConvention('ieee');
        const stages = {};
        const rel = function (x, y) {
            let d = 0, m = 0, nan = 0;
            for (let i = 0; i < x.length; i++) {
                if (Number.isNaN(x[i]) || Number.isNaN(y[i])) { if (Number.isNaN(x[i]) !== Number.isNaN(y[i])) nan++; continue; }
                d = Math.max(d, Math.abs(x[i] - y[i])); m = Math.max(m, Math.abs(x[i]));
            }
            return { max_rel: m > 0 ? d / m : 0, nan_mismatch: nan };
        };
        const cell = function (p, i) {
            const r = Math.fround(Math.sqrt(Math.fround(Math.fround(p[4 * i] * p[4 * i]) + Math.fround(p[4 * i + 1] * p[4 * i + 1]))));
            const c = function (u, W) { const t = Math.fround(u * W); return !(t >= 0) ? 0 : (t >= W ? W - 1 : Math.floor(t)); };
            return c(r, spec.nr) + spec.nr * c(p[4 * i + 2], spec.nz);
        };
        Object.keys(a).forEach(function (key) {
            const st = rel(a[key], b[key]);
            if (/position_A$/.test(key)) {
                let cells = 0, alive = 0;
                for (let i = 0; i < n; i++) {
                    if (cell(a[key], i) !== cell(b[key], i)) cells++;
                    if ((a[key][4 * i + 3] > 0.5) !== (b[key][4 * i + 3] > 0.5)) alive++;
                }
                st.cells_differ = cells; st.alive_differ = alive; st.particles = n;
            }
            if (/moments01$/.test(key)) {
                let touched = 0;
                for (let i = 3; i < a[key].length; i += 4) if ((a[key][i] > 0) !== (b[key][i] > 0)) touched++;
                st.touched_cells_differ = touched;
            }
            stages[key] = st;
        });
        report.scenes[name] = { spec: spec, frames: cfg.frames, stages: stages };
    });
    writeJson('conventions.json', report);
})();

// (9) the reference's dense iterative solver (matrix_webgl.js, SURVEY 8(f) next-4), evaluated in
// software the same way as section 8.  Inputs are float32-representable and stored in the blob.
(function () {
    const quiet = { log: function () {} };       // solve() prints R, C and every iterate
    const chunks = [], cases = {};
    let offset = 0;
    function put(arr) {
        const a = new Float32Array(arr);
        const at = [offset, a.length];
        chunks.push(Buffer.from(a.buffer));
        offset += a.length;
        return at;
    }
    function run(name, cfg) {
        const sw = makeSoftwareGL({ fbo: ['x_guess', 'x_result', 'x_stats', 'R', 'C', 'mv_product'], tex: ['m_set', 'x_set', 'b_set'] });
        const registry = {};
        const extra = { console: quiet, document: {} };
        loadAmd('utilities', registry, extra);
        const mw = loadAmd('matrix_webgl', registry, extra);
        const spec = { n_power: cfg.n_power, webgl: sw.gl };
        if (cfg.relaxation !== undefined) spec.relaxation = cfg.relaxation;
        const eq = mw.makeSORIterative(spec);
        const L = eq.vec_length;
        let s = cfg.seed;
        function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
        const A = [], Aflat = new Float32Array(L * L), b = [], x0 = [];
        for (let r = 0; r < L; r++) {
            A.push([]);
            let off = 0;
            for (let c = 0; c < L; c++) { const v = Math.fround((rnd() - 0.5) * cfg.coupling); A[r].push(v); off += Math.abs(v); }
            A[r][r] = Math.fround(cfg.dominance * off + 1 + rnd());
            for (let c = 0; c < L; c++) Aflat[c + L * r] = A[r][c];
            b.push(Math.fround(2 * rnd() - 1));
            x0.push(Math.fround(cfg.x0 * (rnd() - 0.5)));
        }
        const by = {};
        sw.state.fbos.forEach(function (t) { if (t.name) by[t.name] = t; });
        const out = { n_power: cfg.n_power, relaxation: cfg.relaxation === undefined ? null : cfg.relaxation, vec_length: L,
            vec_height: eq.vec_height, A: put(Aflat), b: put(b), x0: put(x0), calls: [] };
        eq.set_matrix(A).set_b(b).init_vector(x0);
        out.x_after_init = put(by.x_result.array);
        cfg.calls.forEach(function (params) {
            const res = eq.solve(params);
            out.calls.push({ params: params, correlation: Number.isNaN(res.correlation) ? 'NaN' : res.correlation, diff: res.diff, iterations: res.iterations,
                result: put(res.result), x_result: put(by.x_result.array), x_guess: put(by.x_guess.array),
                x_stats: put(by.x_stats.array), R: put(by.R.array), C: put(by.C.array) });
        });
        cases[name] = out;
    }
    run('p1_jacobi', { n_power: 1, seed: 11, coupling: 1.0, dominance: 1.5, x0: 0.0,
        calls: [{ tolerance: 1e-7, max_iterations: 1 }, { tolerance: 1e-7, max_iterations: 1 }, { tolerance: 1e-7, max_iterations: 6 }] });
    run('p2_relaxed', { n_power: 2, relaxation: 0.8, seed: 22, coupling: 0.5, dominance: 1.2, x0: 1.0,
        calls: [{ tolerance: 1e-6, substep: 2, max_iterations: 4 }, { tolerance: 0.5, max_iterations: 50 }] });
    run('p3_jacobi', { n_power: 3, seed: 33, coupling: 0.25, dominance: 2.0, x0: 0.5,
        calls: [{ tolerance: 1e-5, max_iterations: 5 }, { tolerance: 1e-5 }] });
    fs.writeFileSync(path.join(outDir, 'swgl_sor.f32.gz'), zlib.gzipSync(Buffer.concat(chunks), { level: 9 }));
    writeJson('swgl_sor.json', {
        what: "the reference's makeSORIterative (host code and shader strings) evaluated by oracle/swgl.js + glsl_eval.js; arrays are [offset, length] into the float32 blob; A is row-major A[col + L*row]",
        file: 'swgl_sor.f32.gz', cases: cases,
    });
})();

console.log('golden fixtures written to ' + outDir);
