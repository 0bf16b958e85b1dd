#!/usr/bin/env node
/*
 * fusionsim_node.js — the reference demo's scene, driven from Node through the native core.
 *
 * The scene is the one fusionsim.js builds in the browser (fusionsim.js:71-148): a 1 m x 2 m
 * cylinder on a 400 x 800 grid, 160 000 protons started in a 20 cm cube around (0, 0, 1 m), a
 * sink on the outer wall and the two end plates, a source block near the axis, and two opposed
 * 10 MA current loops at the ends (a spindle cusp).  The frame loop is the reference's
 * (fusionsim.js:170-178): step(), density(), show the picture.  Here the picture is written as a
 * binary PGM of the running-average density instead of being drawn on a canvas.
 *
 *   node examples/fusionsim_node.js [--frames 100] [--every 10] [--out DIR] [--seed 1] [--raster 4]
 *
 * --raster b draws density()'s point sprites as a rasteriser with b sub-pixel bits does (4: Chromium's SwiftShader, whose
 * run of the reference this reproduces bit for bit; 8: most desktop GPUs); without it: ideal sprites (INTEGRATION.md 4b).
 *
 * Only the import differs from the reference's controller: require('empic_native.js') in place of
 * the AMD module 'empic' (INTEGRATION.md section 2).
 */
'use strict';
const fs = require('fs');
const path = require('path');
const empic = require(path.join(__dirname, '..', 'fusion-sim_amd', 'js', 'empic_native.js'));

const args = { frames: 100, every: 10, out: null, seed: 1, raster: 0 };
for (let i = 2; i < process.argv.length; i += 2) args[process.argv[i].replace(/^--/, '')] = process.argv[i + 1];
const frames = Number(args.frames), every = Number(args.every);

// a small reproducible generator instead of Math.random (the reference is not reproducible, Q8)
let state = (Number(args.seed) >>> 0) || 1;
function random() { state = (Math.imul(state, 1664525) + 1013904223) >>> 0; return state / 4294967296; }

const nparticles = 160000;
const spec = { radius: 1, height: 2, nr: 400, nz: 800, dt: 2e-9, nparticles: 400,
    particle_mass: 1.67e-27, particle_charge: 1.602e-19 };
if (Number(args.raster)) spec.raster_subpixel_bits = Number(args.raster);
const simulation = empic.makeCylindricalParticlePusher(spec);

const sink = [], source = [], position = [], velocity = [];
for (let i = 0; i < spec.nr; i++) {
    sink.push(new Array(spec.nz).fill(1.0));
    source.push(new Array(spec.nz).fill(0.0));
}
for (let j = 0; j < spec.nz; j++) sink[spec.nr - 1][j] = 0;                       // outer wall
for (let i = 1; i < spec.nr - 1; i++) { sink[i][0] = 0; sink[i][spec.nz - 1] = 0; } // end plates
for (let i = 0; i < 50; i++) for (let j = 350; j < 450; j++) source[i][j] = 1.0;   // source block
for (let p = 0; p < nparticles; p++) {
    position.push([0.2 * (random() - 0.5), 0.2 * (random() - 0.5), 0.2 * (random() - 0.5) + 1]);
    velocity.push([0.002 * (random() - 0.5), 0.002 * (random() - 0.5), 0.002 * (random() - 0.5)]);
}
const entropy = new Float32Array(4 * 1024 * 1024), rand = new Float32Array(4 * nparticles);
for (let k = 0; k < entropy.length; k++) entropy[k] = random();
for (let k = 0; k < rand.length; k++) rand[k] = random();

simulation.set({ position: position, velocity: velocity, sink_mask: sink, source_pdf: source });
simulation.setRandomState({ entropy: entropy, rand: rand });
simulation.addCurrentLoop(0.8, 2.0, -10000000);
simulation.addCurrentLoop(0.8, 0.0, 10000000);
simulation.precalc();
simulation.density();

function writeImage(frame) {
    // programDensity shows 0.5 * n of moments01_avgA (empic.js:1090-1116); 8-bit here
    const rgba = simulation.readDensity();
    const img = Buffer.alloc(spec.nr * spec.nz);
    for (let k = 0; k < spec.nr * spec.nz; k++) {
        const v = 0.5 * rgba[4 * k + 3];
        img[k] = v !== v ? 0 : Math.max(0, Math.min(255, Math.round(255 * v)));
    }
    const file = path.join(args.out, 'density_' + String(frame).padStart(5, '0') + '.pgm');
    fs.writeFileSync(file, Buffer.concat([Buffer.from('P5\n' + spec.nr + ' ' + spec.nz + '\n255\n'), img]));
    return file;
}

if (args.out) fs.mkdirSync(args.out, { recursive: true });
const t0 = Date.now();
let written = 0;
for (let frame = 1; frame <= frames; frame++) {
    simulation.step();
    simulation.density();
    if (args.out && frame % every === 0) { writeImage(frame); written++; }
}
const p = simulation.getParticles();       // waits for the stream
const seconds = (Date.now() - t0) / 1000;
let alive = 0;
for (let k = 0; k < nparticles; k++) alive += p.alive[k];
const dens = simulation.readDensity();
let total = 0;
for (let k = 0; k < spec.nr * spec.nz; k++) if (dens[4 * k + 3] === dens[4 * k + 3]) total += dens[4 * k + 3];
console.log(JSON.stringify({ frames: frames, seconds: seconds, fps: frames / seconds, particles: nparticles, alive: alive,
    images: written, density_sum: total, arch: empic.buildArch(), comm: simulation.commInfo() }));   // comm: { rank, world } as the library sees them
simulation.destroy();
