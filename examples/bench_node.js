#!/usr/bin/env node
/*
 * bench_node.js — the headline frame of bench.py, driven from the host north_star names: JavaScript (Node, N-API).
 *
 * BASELINE configs[1]: 1024 x 1024 (r,z) grid, 1e8 particles, fp32, the reference's generator, uniform Bz = 0.01 T, the
 * sink on the outer wall and the end plates (fusionsim.js:94-112) — SURVEY.md 8(d)'s uniform plasma, uploaded from
 * Float32Arrays through empic_native.js (same factory and method names as the reference's `empic`).  One frame is what
 * bench.py times through ctypes: precalc(); step(); density();  — the controller's loop (fusionsim.js:170-178) is
 * step(); density(); per frame, precalc() once; `--frame reference` times that instead.
 *
 *   node examples/bench_node.js [--side 10000] [--grid 1024] [--steps 20] [--warmup 3] [--frame bench|reference]
 *
 * Prints ONE JSON line: value (particle-updates/s), ms_per_step, upload_ms (set() + setRandomState() of all particles from
 * typed arrays), generate_ms, the library's own kernel timings, node's version.  bench.py runs it as extensions.node_host.
 */
'use strict';
const path = require('path');
const empic = require(path.join(__dirname, '..', 'fusion-sim_amd', 'js', 'empic_native.js'));

const args = { side: 10000, grid: 1024, steps: 20, warmup: 3, frame: 'bench', seed: 0x5EEDF051 };
for (let i = 2; i + 1 < process.argv.length; i += 2) args[process.argv[i].replace(/^--/, '')] = process.argv[i + 1];
const side = Number(args.side), grid = Number(args.grid), steps = Number(args.steps), warmup = Number(args.warmup);
const n = side * side;
const now = () => Number(process.hrtime.bigint()) * 1e-6; // ms

// xorshift128: four 32-bit words of state, one float in [0, 1) per call (Math.random cannot be seeded)
let s0 = (Number(args.seed) >>> 0) || 1, s1 = 0x9E3779B9, s2 = 0x243F6A88, s3 = 0xB7E15162;
function random() {
    let t = s3; t ^= t << 11; t ^= t >>> 8;
    s3 = s2; s2 = s1; s1 = s0;
    t ^= s0; t ^= s0 >>> 19; s0 = t;
    return (t >>> 8) / 16777216;
}

const spec = { radius: 1.0, height: 1.0, nr: grid, nz: grid, dt: 2e-9, nparticles: side, particle_mass: 1.67e-27, particle_charge: 1.602e-19 };
const t_gen = now();
const position = new Float32Array(3 * n), velocity = new Float32Array(3 * n);
for (let p = 0; p < n; p++) {
    const rh = Math.max(Math.sqrt(random()), 1e-6), th = 2 * Math.PI * random();
    position[3 * p] = rh * Math.cos(th) * spec.radius;
    position[3 * p + 1] = rh * Math.sin(th) * spec.radius;
    position[3 * p + 2] = random() * spec.height;
    // isotropic Maxwellian, v_th = 1e-3 c (Box-Muller; the third component from a second pair)
    const a = Math.sqrt(-2 * Math.log(1 - random())), b = 2 * Math.PI * random();
    const c = Math.sqrt(-2 * Math.log(1 - random())), d = 2 * Math.PI * random();
    velocity[3 * p] = 1e-3 * a * Math.cos(b);
    velocity[3 * p + 1] = 1e-3 * a * Math.sin(b);
    velocity[3 * p + 2] = 1e-3 * c * Math.cos(d);
}
const entropy = new Float32Array(4 * 1024 * 1024), rand = new Float32Array(4 * n);
for (let k = 0; k < entropy.length; k++) entropy[k] = random();
for (let k = 0; k < rand.length; k++) rand[k] = random();
const sink = new Float32Array(grid * grid).fill(1.0);       // [i][j] -> i * nz + j (flattenGrid's order)
for (let j = 0; j < grid; j++) sink[(grid - 1) * grid + j] = 0;
for (let i = 1; i < grid - 1; i++) { sink[i * grid] = 0; sink[i * grid + grid - 1] = 0; }
const generate_ms = now() - t_gen;

const simulation = empic.makeCylindricalParticlePusher(spec);
const t_up = now();
simulation.set({ position: position, velocity: velocity, sink_mask: sink, source_pdf: sink });
simulation.setRandomState({ entropy: entropy, rand: rand });
simulation.sync();
const upload_ms = now() - t_up;
simulation.addBZ(0.01);
simulation.precalc();
simulation.sync();
const t_bin = now();
simulation.sort();   // the first binning of the randomly ordered upload belongs to setup, like the upload (bench.py does the same)
simulation.sync();
const first_binning_ms = now() - t_bin;

const reference_frame = args.frame === 'reference';
function frame() {
    if (!reference_frame) simulation.precalc();
    simulation.step();       // fusionsim.js:172
    simulation.density();    // fusionsim.js:174
}
for (let k = 0; k < warmup; k++) frame();
simulation.sync();
simulation.resetStats();
simulation.profile(true);
simulation.sync();
const t0 = now();
for (let k = 0; k < steps; k++) frame();
simulation.sync();
const elapsed_ms = now() - t0;
const st = simulation.stats();
simulation.profile(false);
simulation.destroy();

const out = {
    metric: 'particle-updates/sec (push+deposit+solve)', value: 2.0 * n * steps / (elapsed_ms * 1e-3), unit: 'particle-updates/s',
    n_gpus: 1, steps: steps, warmup: warmup, ms_per_step: elapsed_ms / steps, dtype: 'f32', data: 'synthetic',
    host: 'node ' + process.version + ' through fusionpic_napi.node (N-API) and empic_native.js',
    frame: reference_frame ? 'step(); density();  (fusionsim.js:170-178; precalc() once)' : 'precalc(); step(); density();  (what bench.py times through ctypes)',
    config: { workload: '2D axisymmetric (r,z) ' + grid + 'x' + grid + ' grid, ' + n + ' particles, single species, reference stamp (11x11) deposit, uniform Bz=0.01 T', particles_per_gpu: n, grid: [grid, grid] },
    upload_ms: upload_ms, upload_what: 'set({position, velocity, sink_mask, source_pdf}) + setRandomState({entropy, rand}) from typed arrays: ' + (28 * n + 16777216 + 8 * grid * grid) + ' bytes, then sync()',
    generate_ms: generate_ms, first_binning_ms: first_binning_ms,
    push_avg_launch_ms: st.step_launches ? st.ms_push / st.step_launches : null,
    kernel_ms_per_step: { push: st.ms_push / steps, stamp_normalise_ema: st.ms_stamp / steps, precalc: st.ms_precalc / steps, bin_table_scan: st.ms_sort / steps },
};
console.log(JSON.stringify(out));
