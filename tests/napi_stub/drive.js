/* drive.js — TEST INFRASTRUCTURE (make -C fusion-sim_amd sanitize): every method of the JavaScript host layer
 * (js/empic_native.js, js/matrix_native.js) and every entry point of the N-API addon under them, against the stub
 * libfusionpic.so of this directory, with the addon and the stub built with AddressSanitizer + UBSan.  Right-sized typed
 * arrays go through the shim; WRONG-sized ones, wrong element types, missing arguments and dead handles go to the raw
 * addon, where each must come back as a JavaScript exception (a RangeError / TypeError / Error), never as a native
 * out-of-bounds access — which the sanitizers would report and turn into a non-zero exit.  Also: stepAsync on the worker
 * pool (and the "busy" guard around it), destroy() twice, handles dropped without destroy() and collected (finalisers). */
'use strict';
const path = require('path');
const assert = require('assert');
const empic = require(path.join(__dirname, '..', '..', 'fusion-sim_amd', 'js', 'empic_native.js'));
const matrix = require(path.join(__dirname, '..', '..', 'fusion-sim_amd', 'js', 'matrix_native.js'));
const raw = empic._addon();
assert(/stub/.test(empic.buildArch()), 'this driver must run against the stub library (FUSIONPIC_NAPI_ADDON), got ' + empic.buildArch());

let thrown = 0;
function mustThrow(f, what) {
    let ok = false;
    try { f(); } catch (e) { ok = true; thrown++; assert(e instanceof Error, what); }
    assert(ok, 'no exception: ' + what);
}

// ---------------------------------------------------------------- the (r,z) pusher, both precisions
for (const precision of ['fp32', 'fp64']) {
    const side = 7, n = side * side, nr = 12, nz = 9;
    const Real = precision === 'fp64' ? Float64Array : Float32Array;
    const spec = { radius: 1, height: 2, nr: nr, nz: nz, dt: 1e-9, nparticles: side, particle_mass: 1.67e-27, particle_charge: 1.6e-19, precision: precision };
    const sim = empic.makeCylindricalParticlePusher(spec);
    const nested = (a, b, c) => Array.from({ length: a }, () => Array.from({ length: b }, () => c === 0 ? 1.0 : [0.1, 0.2, 0.3]));
    sim.set({ E: nested(nr, nz, 3), B: new Float64Array(3 * nr * nz), position: Array.from({ length: n }, () => [0.1, 0.2, 0.3]), velocity: new Float32Array(3 * n),
              sink_mask: nested(nr, nz, 0), source_pdf: new Float32Array(nr * nz).fill(1) });
    sim.setRandomState({ entropy: new Float32Array(4 * 1024 * 1024), rand: new Float32Array(4 * n) });
    sim.setRandomState({ entropy: null, rand: Array.from({ length: 4 * n }, () => 0.5) });
    sim.addCurrentLoop(0.8, 2.0, -1e7); sim.addCurrentZ(1e3); sim.addBZ(0.01); sim.addBTheta(0.02);
    mustThrow(() => sim.addSpindleCuspPlasmaField(1, 2, 3), 'spindle');
    sim.precalc(); sim.step(); sim.step(3); sim.substeps(1); sim.density(); sim.deposit(); sim.densityFinish(); sim.sort(); sim.sync();
    for (const name of ['moments', 'norm', 'avg', 'R1', 'R2', 'R3', 'A', 'B', 'E', 'sink', 'inv_cdf']) {
        const g = sim.readGrid(name);
        assert(g instanceof Real && g.length === 4 * (name === 'inv_cdf' ? 512 * 512 : nr * nz), name);
    }
    sim.readDensity(new Real(4 * nr * nz)); sim.readMoments();
    const p = sim.getParticles();
    assert(p.position.length === 3 * n && p.rand.length === 4 * n && p.alive.length === n);
    sim.getParticles({ position: new Real(3 * n), velocity: null, rand: null, alive: new Uint8Array(n) });
    assert(sim.getCells().length === n);
    sim.profile(true); const st = sim.stats(); assert('ms_push' in st && 'step_launches' in st); sim.resetStats();
    sim.commInit(empic.commUniqueId(), 0, 1, true); assert.deepStrictEqual(Object.keys(sim.commInfo()).sort(), ['rank', 'world']); sim.commDestroy();
    sim.saveCheckpoint('/tmp/x.ckpt'); mustThrow(() => sim.loadCheckpoint('/tmp/none.ckpt'), 'loadCheckpoint of nothing');
    // the shim's own length checks
    mustThrow(() => sim.readDensity(new Real(4 * nr * nz - 1)), 'short out buffer (shim)');
    mustThrow(() => sim.set({ position: new Float32Array(3 * n - 3) }), 'short position (shim)');
    mustThrow(() => sim.getCells(new Int32Array(n + 1)), 'long cells (shim)');
    sim.destroy(); sim.destroy();
    mustThrow(() => sim.step(), 'step after destroy');
}

// ---------------------------------------------------------------- the raw addon: what the shim would have stopped
{
    const n = 16, nr = 8, nz = 8;
    const h = raw.create(1, 1, nr, nz, 1e-9, 4, 1e-27, 1e-19, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    mustThrow(() => raw.setParticles(h, new Float32Array(3 * n - 3), null), 'short position');
    mustThrow(() => raw.setParticles(h, new Float32Array(3 * n + 1), null), 'position not a multiple of 3');
    mustThrow(() => raw.setParticles(h, new Int32Array(3 * n), null), 'position of the wrong element type');
    mustThrow(() => raw.setParticles(h, [1, 2, 3], null), 'position not a typed array');
    mustThrow(() => raw.setGrid(h, 0, new Float32Array(3 * nr * nz - 1), nr, nz, 3), 'short grid');
    mustThrow(() => raw.setGrid(h, 0, null, nr, nz, 3), 'null grid');
    mustThrow(() => raw.setRandomState(h, new Float32Array(100), null), 'short entropy');
    mustThrow(() => raw.setRandomState(h, null, new Float32Array(4 * n + 4)), 'long rand');
    mustThrow(() => raw.setRandomState(h, null, new Float64Array(4 * n)), 'rand of the wrong element type');
    mustThrow(() => raw.readGrid(h, 2, new Float32Array(4 * nr * nz - 4)), 'short grid read-back');
    mustThrow(() => raw.readGrid(h, 10, new Float32Array(4 * nr * nz)), 'inv_cdf into a grid-sized buffer');
    mustThrow(() => raw.readGrid(h, 99, new Float32Array(4 * nr * nz)), 'unknown grid');
    mustThrow(() => raw.getParticles(h, new Float32Array(3 * n - 1), null, null, null), 'short particle read-back');
    mustThrow(() => raw.getParticles(h, null, null, new Float32Array(4 * n), new Uint8Array(n - 1)), 'short alive flags');
    mustThrow(() => raw.getCells(h, new Int32Array(n - 1)), 'short cells');
    mustThrow(() => raw.getCells(h, new Float32Array(n)), 'cells of the wrong element type');
    mustThrow(() => raw.step(h), 'missing argument');
    mustThrow(() => raw.step({}, 1), 'not a handle');
    mustThrow(() => raw.commInit(h, new Uint8Array(64), 0, 1, 1), 'short unique id');
    mustThrow(() => raw.commInit(h, empic.commUniqueId(), 3, 2, 1), 'rank outside the world');
    raw.destroy(h);
    mustThrow(() => raw.precalc(h), 'dead handle');
    raw.destroy(h);   // (idempotent, like the shim's destroy())
}

// ---------------------------------------------------------------- the CART3D box (extension keys), electrostatic and full EM
for (const solver of ['poisson_fft', 'yee', 'none']) {
    const n = 50, nx = 8, ny = 4, nz = 16, nodes = nx * ny * nz;
    const box = empic.makeCylindricalParticlePusher({ radius: 1e-2, length_y: 1e-2, height: 2e-2, nr: nx, ny: ny, nz: nz, dt: 1e-12, nparticles: 0, count: n,
        particle_mass: 9.1e-31, particle_charge: -1.6e-19, geometry: 'cart3d', solver: solver, macro_weight: 1e6 });
    const ions = box.addSpecies(1.67e-27, 1.6e-19, 30);
    box.set({ position: new Float32Array(3 * n), velocity: Array.from({ length: n }, () => [0, 0, 0.1]) });
    box.set({ position: new Float64Array(90), velocity: new Float64Array(90) }, ions);
    box.setRange(10, { position: new Float32Array(3 * 5), velocity: new Float32Array(3 * 5) });
    if (solver === 'none') box.set({ E: new Float64Array(3 * nodes) });
    box.addB(0, 0, 0.01); box.addBZ(0.01);
    box.precalc(); box.step(2); box.substeps(3); box.density();
    for (const name of ['E', 'rho', 'phi', 'rho_fixed', 'B', 'edge_E', 'face_B', 'J_fixed']) {
        const f = box.readField(name);
        assert(f.length === nodes * (name === 'rho' || name === 'phi' || name === 'rho_fixed' ? 1 : (name === 'J_fixed' ? 3 : 4)), name);
        assert((f instanceof BigInt64Array) === (name === 'rho_fixed' || name === 'J_fixed'), name);
    }
    assert(box.getParticles(undefined, ions).position.length === 90 && box.getCells(undefined, ions).length === 30);
    assert(box.getRange(3, 7, undefined, 0, 5).velocity.length === 21);
    box.commInit(empic.commUniqueId(), 1, 2); box.domainInit(1, 2, { ghost_planes: 2, migrate_every: 3, distributed_solve: 'interface' });
    box.domainSet({ position: new Float32Array(3 * 20), velocity: new Float32Array(3 * 20) }, 100);
    const got = box.domainGet(); assert(got.n === 25 && got.ids.length === 25 && got.position.length === 75);
    assert(box.domainStats().migrated === 7);
    mustThrow(() => box.readField('nothing'), 'unknown field');
    mustThrow(() => box.getRange(0, 1000), 'range outside the species');
    const h = null; void h;
    box.destroy();
}
{   // raw addon, box handle
    const h = raw.create(1e-2, 2e-2, 8, 16, 1e-12, 0, 9.1e-31, -1.6e-19, 40, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 4, 1e-2, 1, 0, 0);
    mustThrow(() => raw.setParticlesRange(h, 0, 30, new Float32Array(3 * 20), null), 'range beyond the species');
    mustThrow(() => raw.setParticlesRange(h, 5, 0, new Float32Array(3), null), 'unknown species');
    mustThrow(() => raw.getParticlesOf(h, 0, new Float32Array(3 * 39), null), 'short read-back of a species');
    mustThrow(() => raw.getParticlesRange(h, 0, 0, 1, new Float32Array(3 * 50), null), 'read-back range beyond the species');
    mustThrow(() => raw.setField3(h, 0, new Float32Array(3 * 8 * 4 * 16 - 3), 8, 4, 16), 'short field');
    mustThrow(() => raw.readField3(h, 3, new Float32Array(8 * 4 * 16)), 'integer grid into a float array');
    mustThrow(() => raw.readField3(h, 0, new Float32Array(8 * 4 * 16)), 'short field read-back');
    mustThrow(() => raw.domainGetParticles(h, 0, new Float32Array(3 * 40), new Float32Array(3 * 40), new Uint32Array(39)), 'short ids');
    mustThrow(() => raw.domainSetParticles(h, 0, new Float32Array(3 * 10), new Float32Array(3 * 9), 0), 'position and velocity of different lengths');
    // (h is dropped without destroy(): its finaliser runs at the collection below)
}

// ---------------------------------------------------------------- the dense solver
{
    const s = matrix.makeSORIterative({ n_power: 2, relaxation: 0.9 });
    const L = s.vec_length;
    assert(L === 64 && s.vec_height === 4);
    s.set_matrix(new Float32Array(L * L)).set_b(new Float64Array(L)).init_vector(Array.from({ length: L }, () => 0));
    s.mv_product(); s.iterate(3); s.sync();
    const r = s.solve({ tolerance: 1e-6, substep: 2, max_iterations: 10 });
    assert(r.result.length === L && 'correlation' in r);
    assert(s.x_result_tex().read().length === L && s.readVector(2).length === L && s.readIterationMatrix().length === L * L);
    mustThrow(() => raw.sorSet(raw.sorCreate(2, 0, 0, 0), 0, new Float32Array(L * L - 1)), 'short matrix');
    mustThrow(() => raw.sorRead(raw.sorCreate(2, 0, 0, 0), -1, new Float32Array(L)), 'short iteration matrix read-back');
    mustThrow(() => matrix.makeSORIterative({ n_power: 0 }), 'n_power 0');
    s.destroy();
}

// ---------------------------------------------------------------- stepAsync on the worker pool, the guard around it, finalisers
(async function () {
    const sim = empic.makeCylindricalParticlePusher({ radius: 1, height: 1, nr: 8, nz: 8, dt: 1e-9, nparticles: 4, particle_mass: 1e-27, particle_charge: 1e-19 });
    const pending = sim.stepAsync(5);
    mustThrow(() => sim.density(), 'a method while stepAsync runs');
    let second = null;
    try { await sim.stepAsync(1); } catch (e) { second = e; }
    assert(second instanceof Error, 'a second stepAsync while the first runs');
    await pending;
    sim.density();
    await sim.stepAsync();
    sim.destroy();
    for (let k = 0; k < 200; ++k) empic.makeCylindricalParticlePusher({ radius: 1, height: 1, nr: 8, nz: 8, dt: 1e-9, nparticles: 4, particle_mass: 1e-27, particle_charge: 1e-19 });
    if (global.gc) { global.gc(); await new Promise(r => setTimeout(r, 50)); global.gc(); }
    console.log('ok: ' + thrown + ' misuses came back as JavaScript exceptions; no sanitizer report');
})().catch(e => { console.error(e); process.exit(1); });
