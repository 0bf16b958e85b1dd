#!/usr/bin/env node
/*
 * make_golden.js — generates tests/golden/* by IMPORTING the reference's own host
 * JavaScript (read-only, from /root/reference) under Node.  TEST INFRASTRUCTURE.
 *
 *   node oracle/make_golden.js [/root/reference] [tests/golden]
 *
 * What it does: evaluates utilities.js, spindle.js and empic.js through an AMD
 * define() shim, replaces util.webGL with a recorder (no WebGL context exists in
 * the image, so no shader is ever executed), calls the reference factory and its
 * set()/step()/density() methods, and writes ONLY numeric data and call-order
 * metadata: Float32Array contents the reference computed on the host, uniform
 * values it set, and which resource each draw reads and writes.  No reference
 * source text (JS or GLSL) is written anywhere.
 *
 * The fixtures travel to the GPU box; /root/reference does not.
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const zlib = require('zlib');

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

function makeReference() {
    const r = makeRecorder();
    const extra = {
        document: { createElement: function () { return { style: {} }; }, body: { appendChild: function () {} } },
        crypto: { getRandomValues: function (a) { for (let i = 0; i < a.length; i++) a[i] = nextU32(); } },
    };
    const registry = {};
    const util = loadAmd('utilities', registry, extra);
    util.webGL = function () { return r.webgl; };
    const empic = loadAmd('empic', registry, extra);
    return { empic: empic, util: util, rec: r.rec };
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

console.log('golden fixtures written to ' + outDir);
