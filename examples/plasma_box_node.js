#!/usr/bin/env node
/*
 * plasma_box_node.js — the self-consistent box (extension, no reference counterpart) from Node: a cold electron
 * plasma against a neutralising background, displaced by a small sinusoidal perturbation, oscillating at the plasma
 * frequency.  Prints the field energy per frame and the frequency measured from its zero crossings beside
 * omega_p * cos(k dx / 2), the dispersion of the scheme (DESIGN.md 4.4).
 *
 *   node examples/plasma_box_node.js [--grid 32] [--perCell 8] [--frames 120] [--solver poisson_fft|yee]
 *
 * Same surface as the reference's pusher (makeCylindricalParticlePusher -> set / precalc / step), selected by the
 * extension key geometry:'cart3d'.
 */
'use strict';
const path = require('path');
const empic = require(path.join(__dirname, '..', 'fusion-sim_amd', 'js', 'empic_native.js'));

const args = { grid: 32, perCell: 8, frames: 120, solver: 'poisson_fft' };
for (let i = 2; i < process.argv.length; i += 2) args[process.argv[i].replace(/^--/, '')] = process.argv[i + 1];
const g = Number(args.grid), ppc = Number(args.perCell), frames = Number(args.frames);

const eps0 = 8.8541878128e-12, me = 9.109e-31, qe = -1.602e-19, c = 2.998e8;
const dx = 1e-3, L = g * dx;
const yee = args.solver === 'yee';
const dt = yee ? 0.5 * dx / (c * Math.sqrt(3)) : 2e-12;
const wp = 0.05 / dt;                                   // omega_p dt = 0.05
const density = wp * wp * eps0 * me / (qe * qe);
const n = g * g * g * ppc;
const spec = { radius: L, length_y: L, height: L, nr: g, ny: g, nz: g, dt: dt, nparticles: 0, count: n, particle_mass: me,
    particle_charge: qe, geometry: 'cart3d', solver: args.solver, macro_weight: density * L * L * L / n };
const sim = empic.makeCylindricalParticlePusher(spec);

// a regular lattice of particles, displaced along x by a sin(k x): rho = -n0 q a k cos(k x)
const k = 2 * Math.PI / L, amp = 0.02 * dx;
const side = Math.round(Math.cbrt(ppc));
const position = new Float64Array(3 * n), velocity = new Float64Array(3 * n);
let p = 0;
const per = g * side;
for (let a = 0; a < per && p < n; a++) for (let b = 0; b < per && p < n; b++) for (let d = 0; d < per && p < n; d++, p++) {
    const x = (a + 0.5) * L / per;
    position[3 * p] = x + amp * Math.sin(k * x);
    position[3 * p + 1] = (b + 0.5) * L / per;
    position[3 * p + 2] = (d + 0.5) * L / per;
}
sim.set({ position: position.subarray(0, 3 * p), velocity: velocity.subarray(0, 3 * p) });
sim.precalc();

const field = yee ? 'edge_E' : 'E';
function fieldEnergy() {
    const e = sim.readField(field);
    let s = 0;
    for (let i = 0; i < e.length; i += 4) s += e[i] * e[i] + e[i + 1] * e[i + 1] + e[i + 2] * e[i + 2];
    return 0.5 * eps0 * s * dx * dx * dx;
}
// Ex at one node near the crest of the perturbation: its zero crossings give the period
const probe = 4 * (Math.round(g / 4));
let last = sim.readField(field)[probe], crossings = [], t = 0;
for (let frame = 0; frame < frames; frame++) {
    sim.step();                                             // two sub-steps
    t += 2 * dt;
    const ex = sim.readField(field)[probe];
    if ((ex > 0) !== (last > 0)) crossings.push(t - 2 * dt * Math.abs(ex) / (Math.abs(ex) + Math.abs(last)));
    last = ex;
    if (frame % 10 === 0) console.log('frame %d  t = %s ns  field energy %s J', frame, (t * 1e9).toFixed(4), fieldEnergy().toExponential(4));
}
let measured = NaN;
if (crossings.length >= 3) measured = Math.PI * (crossings.length - 1) / (crossings[crossings.length - 1] - crossings[0]);
const expected = wp * Math.cos(k * dx / 2);
// full EM: the exact integer current of the last sub-step (3 int64 per node); its sum over the box is the total
// momentum flux in fixed point, here just shown to be there
let currentNodes = 0;
if (yee) { const J = sim.readField('J_fixed'); for (let i = 0; i < J.length; i++) if (J[i] !== 0n) { currentNodes++; } }
console.log(JSON.stringify({ particles: p, grid: g, solver: args.solver, nonzero_current_entries: currentNodes, omega_measured: measured, omega_scheme: expected, omega_p: wp,
    relative_error: Math.abs(measured - expected) / expected, updates: sim.stats().particle_updates }));
sim.destroy();
