/*
 * empic_native.js — drop-in for the reference's `empic` module on MI355X.
 *
 *   const empic = require('./empic_native.js');
 *   const simulation = empic.makeCylindricalParticlePusher(spec);   // empic.js:30
 *   simulation.set({position, velocity, sink_mask, source_pdf});    // empic.js:1157
 *   simulation.addCurrentLoop(0.8, 2.0, -1e7);                      // empic.js:1352
 *   simulation.precalc();                                           // empic.js:1413
 *   simulation.step(); simulation.density();                        // empic.js:1436, :1471
 *
 * Same factory name, same method names, same argument meaning and the same
 * synchronous `throw new Error(".prop <- ...")` on a bad spec (utilities.js:118-127)
 * as the reference object.  The arithmetic runs in hand-written HIP kernels for
 * gfx950 behind libfusionpic.so; this file only validates, flattens nested arrays
 * into typed arrays, and forwards.  There is no JavaScript or CPU compute path: if
 * the addon or a gfx950 device is missing, the factory throws.
 *
 * What cannot exist outside a browser is `canvas` (empic.js:60).  In its place:
 *   readDensity(out?)   -> Float32Array 4*nr*nz, index 4*(i + j*nr), moments01_avgA
 *   readMoments(out?)   -> moments01;  readGrid(name, out?) for any other texture
 *   getParticles()      -> {position, velocity, rand, alive} in the caller's order
 *   setRandomState({entropy, rand}) -> reproducible runs (the reference seeds from
 *                          window.crypto and Math.random, empic.js:148-173)
 * Extension keys of spec (all optional): precision 'fp32'|'fp64', device (or devices: [d]), count,
 * compat (default true: keep quirk Q1 of empic.js:645), sort_interval, fuse_deposit
 * (default true; 'census' keeps only the tile census and re-binning in step()), rng 'reference'|'counter' + seed (counter = Philox4x32-10 per particle
 * and sub-step instead of the reference's entropy-table generator; not the reference's
 * random stream).
 * shape 'cic' replaces density()'s 11x11 sprite by a bilinear deposit on the four nearest cell centres (extension).
 * raster_subpixel_bits b (1..8) draws density()'s point sprites as a rasteriser with b sub-pixel bits does (snapped window
 * positions, cropped instead of discarded outside the target); 4 reproduces the reference under Chromium's SwiftShader bit
 * for bit, absent / 0 = ideal sprites (include/fusionpic.h).
 * geometry 'cart3d' (+ ny, length_y, solver 'poisson_fft'|'none', macro_weight) selects the self-consistent
 * electrostatic box — an extension with no reference counterpart (include/fusionpic.h): radius, height are
 * then the box lengths along x and z, nr, nz the node counts; same method names, plus addSpecies, addB,
 * readField.  Multi-GPU (one process per GPU): empic.commUniqueId() on rank 0, simulation.commInit(id,
 * rank, world) on every rank; density() then sums the per-cell sums over the ranks inside the library (RCCL).
 */
'use strict';
const path = require('path');

let native = null;
function addon() {
    if (native) return native;
    // (FUSIONPIC_NAPI_ADDON: another build of the same addon — `make -C fusion-sim_amd sanitize` points it at the ASan / UBSan build)
    const file = process.env.FUSIONPIC_NAPI_ADDON || path.join(__dirname, '..', 'lib', 'fusionpic_napi.node');
    try {
        native = require(file);
    } catch (e) {
        throw new Error('fusionpic: cannot load ' + file + ' (' + e.message +
            '); build it with `make -C fusion-sim_amd napi`. There is no CPU fallback.');
    }
    return native;
}

// ---- validate_object / validate_property (utilities.js:11-127), same messages
function validate_property(test, control) {
    if (typeof test === 'undefined') {
        if (!Array.isArray(control) || typeof control[0] !== 'undefined') {
            throw new Error(' <- Non-optional property is undefined!');
        }
    } else if (typeof test !== control) {
        if (Array.isArray(control)) {
            // utilities.js:40-49 is try/finally with no catch: the first defined alternative that does not
            // match throws straight through, later alternatives are never tried (kept, as the reference behaves)
            let failed = true;
            for (let i = 0; failed && i < control.length; i++) {
                if (typeof control[i] !== 'undefined') { validate_property(test, control[i]); failed = false; }
            }
            if (failed) throw new Error(' <- Property does not match any given possible types!');
        } else if (typeof control === 'object' && typeof test === 'object') {
            validate_object(test, control);
        } else {
            throw new Error(' <- Property does not match any given possible types!');
        }
    }
}
function validate_object(test, control) {
    for (const prop in control) {
        try { validate_property(test[prop], control[prop]); } catch (error) { throw new Error('.' + prop + error.message); }
    }
}

const GRID_IN = { E: 0, B: 1, sink_mask: 2, source_pdf: 3 };
const GRID_OUT = { moments: 0, norm: 1, avg: 2, R1: 3, R2: 4, R3: 5, A: 6, B: 7, E: 8, sink: 9, inv_cdf: 10 };

function isFloatArray(a) { return a instanceof Float32Array || a instanceof Float64Array; }

// a caller-supplied output buffer must be exactly as long as what the native call writes
function checkLength(buf, want, name) {
    if (buf !== undefined && buf !== null && buf.length !== want) {
        throw new RangeError('.' + name + ' <- expected a typed array of ' + want + ' elements, got ' + buf.length);
    }
    return buf;
}

// value[i][j][k] or value[i][j] -> Float64Array (JavaScript numbers are doubles)
function flattenGrid(value, nr, nz, ncomp, name) {
    if (isFloatArray(value)) {
        if (value.length !== nr * nz * ncomp) throw new Error('.' + name + ' <- expected ' + (nr * nz * ncomp) + ' elements');
        return value;
    }
    if (!Array.isArray(value) || value.length < nr) throw new Error('.' + name + ' <- expected [' + nr + '][' + nz + ']' + (ncomp > 1 ? '[3]' : ''));
    const out = new Float64Array(nr * nz * ncomp);
    for (let i = 0; i < nr; i++) {
        const row = value[i];
        for (let j = 0; j < nz; j++) {
            if (ncomp === 1) out[i * nz + j] = row[j];
            else for (let k = 0; k < ncomp; k++) out[(i * nz + j) * ncomp + k] = row[j][k];
        }
    }
    return out;
}

function flattenParticles(value, n, name) {
    if (isFloatArray(value)) {
        if (value.length !== 3 * n) throw new Error('.' + name + ' <- expected ' + (3 * n) + ' elements');
        return value;
    }
    if (!Array.isArray(value) || value.length < n) throw new Error('.' + name + ' <- expected [' + n + '][3]');
    const out = new Float64Array(3 * n);
    for (let p = 0; p < n; p++) { out[3 * p] = value[p][0]; out[3 * p + 1] = value[p][1]; out[3 * p + 2] = value[p][2]; }
    return out;
}

// stepAsync(ncalls) -> Promise (SURVEY 8(b), threading): the step runs on a worker thread of Node's pool; the handle is
// not thread-safe, so every other method of the simulation throws until the promise has settled.
function addStepAsync(out, lib, handle) {
    let busy = false;
    for (const name of Object.keys(out)) {
        const f = out[name];
        if (typeof f !== 'function') continue;
        out[name] = function () {
            if (busy) throw new Error('.' + name + ' <- a stepAsync() of this simulation is still running');
            return f.apply(this, arguments);
        };
    }
    out.stepAsync = function (ncalls) {
        if (busy) return Promise.reject(new Error('.stepAsync <- a stepAsync() of this simulation is still running'));
        busy = true;
        const done = () => { busy = false; };
        return lib.stepAsync(handle(), ncalls === undefined ? 1 : ncalls).then(v => { done(); return v; }, e => { done(); throw e; });
    };
    return out;
}

const FIELD3 = { E: 0, rho: 1, phi: 2, rho_fixed: 3, B: 4, edge_E: 5, face_B: 6, J_fixed: 7 };   // B, edge_E, face_B, J_fixed: full EM (solver 'yee')

// spec.geometry === 'cart3d': the electrostatic box behind the same method names
function makeBox(spec, lib) {
    validate_object(spec, { ny: 'number', length_y: 'number', solver: [, 'string'], macro_weight: [, 'number'] });
    if (spec.solver !== undefined && ['poisson_fft', 'none', 'yee'].indexOf(spec.solver) < 0) throw new Error(".solver <- must be 'poisson_fft', 'yee' or 'none'");
    const fp64 = spec.precision === 'fp64';
    const n0 = spec.count ? spec.count : spec.nparticles * spec.nparticles;
    let h = lib.create(spec.radius, spec.height, spec.nr, spec.nz, spec.dt, spec.nparticles, spec.particle_mass, spec.particle_charge,
        spec.count || 0, fp64 ? 1 : 0, spec.device || 0, 0, spec.sort_interval || 0, 0, 0, 0, 0,
        1, spec.solver === 'none' ? 0 : (spec.solver === 'yee' ? 2 : 1), spec.ny, spec.length_y, spec.macro_weight === undefined ? 1 : spec.macro_weight, 0, 0);
    const nx = spec.nr, ny = spec.ny, nz = spec.nz, nodes = nx * ny * nz;
    const counts = [n0];
    const Real = fp64 ? Float64Array : Float32Array;
    const out = {};
    out.addSpecies = function (mass, charge, count) { const s = lib.addSpecies(h, mass, charge, count); counts.push(count); return s; };
    out.set = function (value, species) {                                      // empic.js:1157: position [N][3] m, velocity [N][3] in c
        const sp = species || 0, n = counts[sp];
        if (value.position) lib.setParticlesRange(h, sp, 0, flattenParticles(value.position, n, 'position'), null);
        if (value.velocity) lib.setParticlesRange(h, sp, 0, null, flattenParticles(value.velocity, n, 'velocity'));
        if (value.E) {
            let e = value.E;
            if (!isFloatArray(e)) {                                           // value[i][j][k][3]
                e = new Float64Array(3 * nodes);
                for (let i = 0; i < nx; i++) for (let j = 0; j < ny; j++) for (let k = 0; k < nz; k++)
                    for (let c = 0; c < 3; c++) e[3 * ((i * ny + j) * nz + k) + c] = value.E[i][j][k][c];
            }
            lib.setField3(h, FIELD3.E, checkLength(e, 3 * nodes, 'E'), nx, ny, nz);
        }
    };
    out.setRange = function (first, value, species) { lib.setParticlesRange(h, species || 0, first, value.position || null, value.velocity || null); };
    out.addBZ = function (Bz) { lib.addBZ(h, Bz); };                           // empic.js:1391
    out.addB = function (bx, by, bz) { lib.addB(h, bx, by, bz); };
    out.precalc = function () { lib.precalc(h); };                             // empic.js:1413: fields <- particles (deposit + solve)
    out.step = function (ncalls) { lib.step(h, ncalls === undefined ? 1 : ncalls); };  // empic.js:1436: 2 sub-steps
    out.substeps = function (n) { lib.substeps(h, n === undefined ? 1 : n); };         // extension: single sub-steps, step(k) = substeps(2 k)
    out.density = function () { lib.density(h); };                             // empic.js:1471: the charge density is always current
    out.readField = function (name, buf) {
        if (!(name in FIELD3)) throw new Error('.name <- unknown field ' + name);
        const len = nodes * (name === 'rho' || name === 'phi' || name === 'rho_fixed' ? 1 : (name === 'J_fixed' ? 3 : 4));
        const fresh = name === 'rho_fixed' || name === 'J_fixed' ? new BigInt64Array(len) : new Real(len);   // the exact integer grids
        return lib.readField3(h, FIELD3[name], checkLength(buf, len, 'out') || fresh);
    };
    out.getParticles = function (into, species) {
        const sp = species || 0, n = counts[sp];
        const r = into || { position: new Real(3 * n), velocity: new Real(3 * n) };
        checkLength(r.position, 3 * n, 'position'); checkLength(r.velocity, 3 * n, 'velocity');
        lib.getParticlesOf(h, sp, r.position || null, r.velocity || null);
        return r;
    };
    out.getCells = function (buf, species) { const n = counts[species || 0]; return lib.getCellsOf(h, species || 0, checkLength(buf, n, 'cells') || new Int32Array(n)); };
    // the mirror of setRange, and a sampled read-back: the caller's particles first, first + stride, ... (count of them;
    // stride 1 = the range [first, first + count)); 2e9 particles do not fit one typed array
    out.getRange = function (first, count, into, species, stride) {
        const r = into || { position: new Real(3 * count), velocity: new Real(3 * count) };
        checkLength(r.position, 3 * count, 'position'); checkLength(r.velocity, 3 * count, 'velocity');
        lib.getParticlesRange(h, species || 0, first, stride === undefined ? 1 : stride, r.position || null, r.velocity || null);
        return r;
    };
    out.commInit = function (id, rank, world, overlap) { lib.commInit(h, id, rank, world, overlap === false ? 0 : 1); };
    out.commDestroy = function () { lib.commDestroy(h); };
    out.commInfo = function () { return lib.commInfo(h); };                  // { rank, world } as the library sees them
    // z-slab decomposition (this process = rank `rank` of `world`, after commInit with the same numbers): the rank's
    // particles arrive through domainSet with their global indices; step() then exchanges ghost planes, halos and
    // migrating particles with the neighbours inside the library
    out.domainInit = function (rank, world, options) {
        const o = options || {};
        lib.domainInit(h, rank, world, o.ghost_planes === undefined ? 2 : o.ghost_planes, o.migrate_every === undefined ? 4 : o.migrate_every,
                       o.distributed_solve === 'interface' || o.distributed_solve === 2 ? 2 : (o.distributed_solve ? 1 : 0));
    };
    out.domainSet = function (value, firstId, species) {
        const sp = species || 0;
        const p = isFloatArray(value.position) ? value.position : flattenParticles(value.position, value.position.length, 'position');
        const v = isFloatArray(value.velocity) ? value.velocity : flattenParticles(value.velocity, value.velocity.length, 'velocity');
        lib.domainSetParticles(h, sp, p, v, firstId || 0);
    };
    out.domainGet = function (species) {
        const sp = species || 0, cap = counts[sp];
        const position = new Real(3 * cap), velocity = new Real(3 * cap), ids = new Uint32Array(cap);
        const n = lib.domainGetParticles(h, sp, position, velocity, ids);
        return { n: n, position: position.subarray(0, 3 * n), velocity: velocity.subarray(0, 3 * n), ids: ids.subarray(0, n) };
    };
    out.domainStats = function () { return lib.domainStats(h); };
    out.saveCheckpoint = function (path) { lib.saveCheckpoint(h, String(path)); };   // fpic_save_checkpoint: particles of every species + fields
    out.loadCheckpoint = function (path) { lib.loadCheckpoint(h, String(path)); };
    out.sort = function () { lib.sort(h); };
    out.sync = function () { lib.sync(h); };
    out.profile = function (on) { lib.profile(h, on ? 1 : 0); };
    out.stats = function () { return lib.getStats(h); };
    out.resetStats = function () { lib.resetStats(h); };
    out.destroy = function () { if (h) { lib.destroy(h); h = null; } };
    out.nparticles = n0;
    return addStepAsync(out, lib, () => h);
}

exports.makeCylindricalParticlePusher = function (spec) {
    validate_object(spec, {           // empic.js:31-41
        radius: 'number', height: 'number', nr: 'number', nz: 'number', dt: 'number',
        nparticles: 'number', particle_mass: 'number', particle_charge: 'number',
        precision: [, 'string'], device: [, 'number'], count: [, 'number'], compat: [, 'boolean'],
        sort_interval: [, 'number'], rng: [, 'string'], seed: [, 'number'], geometry: [, 'string'], shape: [, 'string'], raster_subpixel_bits: [, 'number'],
    });
    if (spec.devices !== undefined) {   // SURVEY 8(b) extension key: one process drives one GPU here; N GPUs are N processes (commInit)
        if (!Array.isArray(spec.devices) || spec.devices.length !== 1 || typeof spec.devices[0] !== 'number') {
            throw new Error(".devices <- one process drives one GPU: name one device and start one process per GPU (commUniqueId / commInit)");
        }
        spec = Object.assign({}, spec, { device: spec.devices[0] });
    }
    if (spec.geometry !== undefined && spec.geometry !== 'cyl_rz' && spec.geometry !== 'cart3d') throw new Error(".geometry <- must be 'cyl_rz' or 'cart3d'");
    if (spec.shape !== undefined && spec.shape !== 'ref11' && spec.shape !== 'cic') throw new Error(".shape <- must be 'ref11' or 'cic'");
    if (spec.geometry === 'cart3d') return makeBox(spec, addon());
    // two admissible types: checked by hand, the reference's validator stops at the first alternative
    if (spec.fuse_deposit !== undefined && typeof spec.fuse_deposit !== 'boolean' && spec.fuse_deposit !== 'census') {
        throw new Error(".fuse_deposit <- must be true, false or 'census'");
    }
    const n = spec.count ? spec.count : spec.nparticles * spec.nparticles;   // empic.js:107-109
    const fp64 = spec.precision === 'fp64';
    if (spec.precision !== undefined && spec.precision !== 'fp32' && spec.precision !== 'fp64') {
        throw new Error(".precision <- must be 'fp32' or 'fp64'");
    }
    if (spec.rng !== undefined && spec.rng !== 'reference' && spec.rng !== 'counter') {
        throw new Error(".rng <- must be 'reference' or 'counter'");
    }
    const seed = spec.seed || 0;
    const lib = addon();
    let h = lib.create(spec.radius, spec.height, spec.nr, spec.nz, spec.dt, spec.nparticles, spec.particle_mass,
        spec.particle_charge, spec.count || 0, fp64 ? 1 : 0, spec.device || 0, spec.compat === false ? 1 : 0,
        spec.sort_interval || 0, spec.fuse_deposit === 'census' ? 2 : (spec.fuse_deposit === false ? 1 : 0), spec.rng === 'counter' ? 1 : 0,
        seed % 4294967296, Math.floor(seed / 4294967296) % 4294967296, 0, 0, 0, 0, 0, spec.shape === 'cic' ? 1 : 0, spec.raster_subpixel_bits || 0);
    const nr = spec.nr, nz = spec.nz;
    const Real = fp64 ? Float64Array : Float32Array;
    const out = {};

    out.set = function (value) {                                              // empic.js:1157-1350
        if (value.E) lib.setGrid(h, GRID_IN.E, flattenGrid(value.E, nr, nz, 3, 'E'), nr, nz, 3);
        if (value.B) lib.setGrid(h, GRID_IN.B, flattenGrid(value.B, nr, nz, 3, 'B'), nr, nz, 3);
        if (value.position) lib.setParticles(h, flattenParticles(value.position, n, 'position'), null);
        if (value.velocity) lib.setParticles(h, null, flattenParticles(value.velocity, n, 'velocity'));
        if (value.sink_mask) lib.setGrid(h, GRID_IN.sink_mask, flattenGrid(value.sink_mask, nr, nz, 1, 'sink_mask'), nr, nz, 1);
        if (value.source_pdf) lib.setGrid(h, GRID_IN.source_pdf, flattenGrid(value.source_pdf, nr, nz, 1, 'source_pdf'), nr, nz, 1);
    };
    out.addCurrentLoop = function (r, z, I) { lib.addCurrentLoop(h, r, z, I); };   // empic.js:1352
    out.addCurrentZ = function (I) { lib.addCurrentZ(h, I); };                     // empic.js:1380
    out.addBZ = function (Bz) { lib.addBZ(h, Bz); };                               // empic.js:1391
    out.addBTheta = function (Btheta) { lib.addBTheta(h, Btheta); };               // empic.js:1402
    out.addSpindleCuspPlasmaField = function () {                                 // empic.js:1369
        // the reference's implementation stops at undefined symbols (spindle.js:328, :624, :643)
        throw new Error('addSpindleCuspPlasmaField is not functional in the reference (spindle.js:328)');
    };
    out.precalc = function () { lib.precalc(h); };                                 // empic.js:1413
    out.step = function (ncalls) { lib.step(h, ncalls === undefined ? 1 : ncalls); };  // empic.js:1436
    out.substeps = function (n) { lib.substeps(h, n === undefined ? 1 : n); };         // extension: single sub-steps, step(k) = substeps(2 k)
    out.density = function () { lib.density(h); };                                 // empic.js:1471

    // ---- stand-ins for `canvas` and the reference's unseeded randomness
    out.deposit = function () { lib.deposit(h); };
    out.densityFinish = function () { lib.densityFinish(h); };
    out.readGrid = function (name, buf) {
        if (!(name in GRID_OUT)) throw new Error('.name <- unknown grid ' + name);
        const cells = name === 'inv_cdf' ? 512 * 512 : nr * nz;
        return lib.readGrid(h, GRID_OUT[name], checkLength(buf, 4 * cells, 'out') || new Real(4 * cells));
    };
    out.readDensity = function (buf) { return out.readGrid('avg', buf); };
    out.readMoments = function (buf) { return out.readGrid('moments', buf); };
    out.getParticles = function (into) {
        const r = into || { position: new Real(3 * n), velocity: new Real(3 * n), rand: new Float32Array(4 * n), alive: new Uint8Array(n) };
        checkLength(r.position, 3 * n, 'position'); checkLength(r.velocity, 3 * n, 'velocity');
        checkLength(r.rand, 4 * n, 'rand'); checkLength(r.alive, n, 'alive');
        lib.getParticles(h, r.position || null, r.velocity || null, r.rand || null, r.alive || null);
        return r;
    };
    out.getCells = function (buf) { return lib.getCells(h, checkLength(buf, n, 'cells') || new Int32Array(n)); };
    out.setRandomState = function (state) {
        const f32 = function (a) { return a === undefined || a === null ? null : (a instanceof Float32Array ? a : Float32Array.from(a)); };
        lib.setRandomState(h, checkLength(f32(state.entropy), 4 * 1024 * 1024, 'entropy'), checkLength(f32(state.rand), 4 * n, 'rand'));
    };
    out.commInit = function (id, rank, world, overlap) { lib.commInit(h, id, rank, world, overlap === false ? 0 : 1); };
    out.commDestroy = function () { lib.commDestroy(h); };
    out.commInfo = function () { return lib.commInfo(h); };                  // { rank, world } as the library sees them
    out.saveCheckpoint = function (path) { lib.saveCheckpoint(h, String(path)); };
    out.loadCheckpoint = function (path) { lib.loadCheckpoint(h, String(path)); };
    out.sort = function () { lib.sort(h); };
    out.sync = function () { lib.sync(h); };
    out.profile = function (on) { lib.profile(h, on ? 1 : 0); };
    out.stats = function () { return lib.getStats(h); };
    out.resetStats = function () { lib.resetStats(h); };
    out.destroy = function () { if (h) { lib.destroy(h); h = null; } };
    out.nparticles = n;
    return addStepAsync(out, lib, () => h);
};

exports.validate_object = validate_object;
exports._addon = addon; // shared with matrix_native.js
exports.buildArch = function () { return addon().buildArch(); };
exports.commUniqueId = function () { return addon().commUniqueId(); };   // rank 0; hand the 128 bytes to every rank
