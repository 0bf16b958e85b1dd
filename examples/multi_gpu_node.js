#!/usr/bin/env node
/*
 * multi_gpu_node.js — several GPUs from the JavaScript host (INTEGRATION.md 4a): ONE Node process per GPU, the library's own
 * RCCL communicator, nothing but 128 bytes handed from rank 0 to every rank.
 *
 *   node examples/multi_gpu_node.js --ranks N [--mode rz|box] [--frames 3] [--one-device] [--check]
 *
 * The parent starts N children (child_process.fork; rank r drives device r, or device 0 with --one-device), relays the unique
 * id rank 0 makes (empic.commUniqueId -> simulation.commInit) and collects what every rank ends with.
 *   --mode rz   the reference's pusher, particles sharded by index range, the grid tables replicated; density() sums the
 *               per-cell sums over the ranks inside the library (one all-reduce per frame, beside the next step()): every rank
 *               ends with the density of the WHOLE population.
 *   --mode box  the self-consistent box (geometry 'cart3d', extension) as z-slabs: commInit + domainInit + domainSet, then the
 *               unchanged frame loop; ghost planes, the decomposed Poisson solve and migrating particles are exchanged inside
 *               the library.
 *   --check     the parent also runs the same scene on ONE handle and compares: the ranks' particles bit for bit (both modes),
 *               the density equal on all ranks bit for bit and equal to one handle's up to the summation order (rz).
 * With --one-device the ranks share device 0, which the real RCCL refuses: bind a stand-in through FPIC_RCCL_LIBRARY
 * (tests/fake_rccl/libfakerccl_shm.so; tests/test_gpu_rccl_processes.py runs exactly that).  On a multi-GPU node no switch is
 * needed: the library binds the RCCL of the ROCm installation.
 */
'use strict';
const path = require('path');
const { fork } = require('child_process');
const crypto = require('crypto');
const empic = require(path.join(__dirname, '..', 'fusion-sim_amd', 'js', 'empic_native.js'));

const args = { ranks: 2, mode: 'rz', frames: 3 };
for (let i = 2; i < process.argv.length; i++) {
    const a = process.argv[i];
    if (a === '--one-device' || a === '--check' || a === '--single') args[a.slice(2)] = true;
    else if (a.startsWith('--')) args[a.slice(2)] = process.argv[++i];
}
const world = Number(args.ranks), frames = Number(args.frames), mode = args.mode;
const sortWorld = Number(args['sort-world'] || world);   // (a one-handle run of the box keeps the ranks' particle numbering)

// ---- the scene: every process builds it from the same seed and takes its share
function lcg(seed) { let s = seed >>> 0; return () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; }; }
function sceneRz() {
    const nr = 96, nz = 80, n = 60000, rnd = lcg(5);
    const spec = { radius: 1.0, height: 2.0, nr: nr, nz: nz, dt: 2e-9, nparticles: 0, count: n, particle_mass: 1.67e-27, particle_charge: 1.602e-19 };
    const position = new Float32Array(3 * n), velocity = new Float32Array(3 * n);
    for (let p = 0; p < n; p++) {
        const rh = Math.max(Math.sqrt(rnd()), 1e-6), th = 2 * Math.PI * rnd();
        position[3 * p] = rh * Math.cos(th); position[3 * p + 1] = rh * Math.sin(th); position[3 * p + 2] = 2.0 * rnd();
        for (let c = 0; c < 3; c++) velocity[3 * p + c] = 5e-3 * (2 * rnd() - 1);
    }
    const entropy = new Float32Array(4 * 1024 * 1024), rand = new Float32Array(4 * n);
    for (let k = 0; k < entropy.length; k++) entropy[k] = rnd();
    for (let k = 0; k < rand.length; k++) rand[k] = rnd();
    const sink = new Float32Array(nr * nz).fill(1);
    for (let j = 0; j < nz; j++) sink[(nr - 1) * nz + j] = 0;
    for (let i = 1; i < nr - 1; i++) { sink[i * nz] = 0; sink[i * nz + nz - 1] = 0; }
    return { spec, n, position, velocity, entropy, rand, sink };
}
function sceneBox() {
    const shape = [16, 16, 32], n = 20000, rnd = lcg(7), L = shape.map(s => 1e-3 * s);
    const spec = { radius: L[0], length_y: L[1], height: L[2], nr: shape[0], ny: shape[1], nz: shape[2], dt: 5e-12, nparticles: 0, count: n,
        particle_mass: 9.109e-31, particle_charge: -1.602e-19, geometry: 'cart3d', solver: 'poisson_fft', macro_weight: 1e15 * L[0] * L[1] * L[2] / n };
    const pos = [], vz = Math.min(0.7 * 2 * 1e-3 / (2 * spec.dt * 2.998e8), 0.9);
    for (let p = 0; p < n; p++) pos.push([rnd() * L[0], rnd() * L[1], rnd() * L[2], 0.05 * (2 * rnd() - 1), 0.05 * (2 * rnd() - 1), vz * (2 * rnd() - 1)]);
    // a rank's initial particles are those of its slab: sorted by owner, global index = position in that order
    const nzl = shape[2] / sortWorld;
    const owner = p => Math.floor(Math.floor(p[2] / L[2] * shape[2]) / nzl);
    const order = pos.map((p, i) => i).sort((a, b) => owner(pos[a]) - owner(pos[b]) || a - b);
    const position = new Float64Array(3 * n), velocity = new Float64Array(3 * n), first = new Array(sortWorld + 1).fill(0);
    order.forEach((src, k) => {
        for (let c = 0; c < 3; c++) { position[3 * k + c] = pos[src][c]; velocity[3 * k + c] = pos[src][3 + c]; }
        first[owner(pos[src]) + 1]++;
    });
    for (let r = 0; r < sortWorld; r++) first[r + 1] += first[r];
    return { spec, n, position, velocity, first };
}
const sha = a => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex');
// typed arrays travel between the processes as base64 of their bytes (JSON has no NaN)
const enc = a => Buffer.from(a.buffer, a.byteOffset, a.byteLength).toString('base64');
const dec = (s, Type) => { const b = Buffer.from(s, 'base64'); return new Type(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };

function runRank(rank, id) {
    const device = args['one-device'] ? 0 : rank;
    if (mode === 'rz') {
        const sc = sceneRz(), lo = Math.floor(sc.n * rank / world), hi = Math.floor(sc.n * (rank + 1) / world);
        const sim = empic.makeCylindricalParticlePusher(Object.assign({}, sc.spec, { count: hi - lo, device: device }));
        sim.set({ position: sc.position.subarray(3 * lo, 3 * hi), velocity: sc.velocity.subarray(3 * lo, 3 * hi), sink_mask: sc.sink, source_pdf: sc.sink });
        sim.setRandomState({ entropy: sc.entropy, rand: sc.rand.subarray(4 * lo, 4 * hi) });
        sim.addBZ(0.02); sim.precalc();
        if (world > 1 || id) sim.commInit(id, rank, world);
        for (let f = 0; f < frames; f++) { sim.step(); sim.density(); }
        const density = sim.readDensity(), p = sim.getParticles();
        const info = world > 1 || id ? sim.commInfo() : { rank: 0, world: 1 };
        sim.destroy();
        return { rank, lo, hi, info, density_sha: sha(density), density: enc(density), position_sha: sha(p.position), velocity_sha: sha(p.velocity), position: enc(p.position) };
    }
    const sc = sceneBox();
    const sim = empic.makeCylindricalParticlePusher(Object.assign({}, sc.spec, { count: world > 1 ? 3 * sc.n : sc.n, device: device }));
    if (world > 1) {
        sim.commInit(id, rank, world);
        sim.domainInit(rank, world, { ghost_planes: 2, migrate_every: 2, distributed_solve: true });
        const a = sc.first[rank], b = sc.first[rank + 1];
        sim.domainSet({ position: sc.position.subarray(3 * a, 3 * b), velocity: sc.velocity.subarray(3 * a, 3 * b) }, a);
    } else {
        sim.set({ position: sc.position, velocity: sc.velocity });
    }
    sim.precalc();
    for (let f = 0; f < frames; f++) sim.step();
    let out;
    if (world > 1) {
        const g = sim.domainGet(), st = sim.domainStats();
        out = { rank, n: g.n, ids: enc(g.ids), position: enc(g.position), velocity: enc(g.velocity), migrated: st.migrated, lost: st.lost, info: sim.commInfo() };
    } else {
        const p = sim.getParticles();
        out = { rank, n: sc.n, position: enc(p.position), velocity: enc(p.velocity) };
    }
    sim.destroy();
    return out;
}

if (args.single) {
    // ---- ONE handle holding everything (what --check compares with): prints its result and leaves
    console.log(JSON.stringify(runRank(0, null)));
} else if (process.env.FPIC_NODE_RANK !== undefined) {
    // ---- a rank: make or receive the id, run, report
    const rank = Number(process.env.FPIC_NODE_RANK);
    const go = id => { process.send({ type: 'result', result: runRank(rank, id) }, () => process.exit(0)); };
    if (rank === 0) {
        const id = empic.commUniqueId();                                  // fpic_comm_unique_id
        process.send({ type: 'id', id: Buffer.from(id).toString('base64') });
        go(id);
    } else {
        process.on('message', m => { if (m.type === 'id') go(new Uint8Array(Buffer.from(m.id, 'base64'))); });
    }
} else {
    // ---- the parent: start the ranks, relay the id, collect
    const kids = [], results = new Array(world);
    let left = world, failed = false;
    for (let r = 0; r < world; r++) {
        const k = fork(__filename, process.argv.slice(2), { env: Object.assign({}, process.env, { FPIC_NODE_RANK: String(r) }) });
        kids.push(k);
        k.on('message', m => {
            if (m.type === 'id') kids.forEach((o, q) => { if (q !== 0) o.send(m); });
            else if (m.type === 'result') results[m.result.rank] = m.result;
        });
        k.on('exit', code => { if (code !== 0) failed = true; if (--left === 0) finish(); });
    }
    const watchdog = setTimeout(() => { console.error('a rank did not finish'); kids.forEach(k => k.kill()); process.exit(2); }, 240000);
    function finish() {
        clearTimeout(watchdog);
        if (failed || results.some(r => !r)) { console.log(JSON.stringify({ ok: false, error: 'a rank failed' })); process.exit(1); }
        const out = { ok: true, mode: mode, world: world, frames: frames, comm: results.map(r => r.info) };
        if (mode === 'rz') {
            out.ranks_agree = results.every(r => r.density_sha === results[0].density_sha);
            out.ok = out.ranks_agree && results.every(r => r.info.world === world);
        } else {
            out.held = results.map(r => r.n); out.migrated = results.reduce((s, r) => s + r.migrated, 0); out.lost = results.reduce((s, r) => s + r.lost, 0);
            out.ok = out.lost === 0 && out.migrated > 0;
        }
        if (args.check) {
            // ONE handle holding everything, in a process of its own
            const one = require('child_process').spawnSync(process.execPath, [__filename, '--ranks', '1', '--sort-world', String(world), '--mode', mode, '--frames', String(frames), '--one-device', '--single'],
                { env: process.env, maxBuffer: 1 << 30 });
            if (one.status !== 0) { console.log(JSON.stringify({ ok: false, error: 'the one-handle run failed: ' + one.stderr.toString().slice(-300) })); process.exit(1); }
            const ref = JSON.parse(one.stdout.toString().trim().split('\n').pop());
            if (mode === 'rz') {
                const want = dec(ref.density, Float32Array), got = dec(results[0].density, Float32Array), refPos = dec(ref.position, Float32Array);
                let top = 0, err = 0, nan_ok = true;
                want.forEach(v => { if (v === v) top = Math.max(top, Math.abs(v)); });
                got.forEach((v, i) => { const w = want[i]; if ((v !== v) !== (w !== w)) nan_ok = false; else if (v === v) err = Math.max(err, Math.abs(v - w)); });
                out.density_rel_err = err / top; out.nan_sites_agree = nan_ok;
                out.particles_same = results.every(r => sha(refPos.subarray(3 * r.lo, 3 * r.hi)) === r.position_sha);
                out.ok = out.ok && nan_ok && out.density_rel_err <= 1e-5 && out.particles_same;
            } else {
                const refPos = dec(ref.position, Float32Array), refVel = dec(ref.velocity, Float32Array);
                const pos = new Float32Array(3 * ref.n).fill(NaN), vel = new Float32Array(3 * ref.n).fill(NaN);
                let seen = 0;
                results.forEach(r => {
                    const ids = dec(r.ids, Uint32Array), p = dec(r.position, Float32Array), v = dec(r.velocity, Float32Array);
                    ids.forEach((id, k) => { for (let c = 0; c < 3; c++) { pos[3 * id + c] = p[3 * k + c]; vel[3 * id + c] = v[3 * k + c]; } seen++; });
                });
                out.every_particle_once = seen === ref.n;
                out.particles_same = out.every_particle_once && sha(pos) === sha(refPos) && sha(vel) === sha(refVel);
                out.ok = out.ok && out.particles_same;
            }
        }
        console.log(JSON.stringify(out));
        process.exit(out.ok ? 0 : 1);
    }
}
