/*
 * pic_oracle.js — plain-JavaScript twin of the CPU oracle (oracle/pic_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY.  It is the "JS/CPU path" that bench.py times on the GPU
 * box's host cores next to the HIP path, and an independent second restatement of
 * the reference's shaders: tests check that it agrees with the C oracle bit for bit.
 * It is the build's own code, not the reference's — the reference has no CPU engine
 * (its arithmetic is GLSL, empic.js:692-1035).  Citations are file:line under
 * /root/reference/public/javascripts/.
 *
 * fp32 is emulated with Math.fround after every operation (exact for + - * / sqrt
 * of float operands); precision 'fp64' makes fround the identity.
 *
 * Textures are flat RGBA arrays, texel (i,j) of a W-wide texture at 4*(i + j*W).
 */
'use strict';

const N_ENTROPY = 1024, N_CDF = 512, NSHAPE = 11;

function makeOracle(spec, opts) {
    opts = opts || {};
    const fp64 = opts.precision === 'fp64';
    const f = fp64 ? function (x) { return x; } : Math.fround;
    const Real = fp64 ? Float64Array : Float32Array;
    const nr = spec.nr, nz = spec.nz;
    const n = opts.count || spec.nparticles * spec.nparticles;
    // empic.js:27, :44-46, :852; literals N(x) = toFixed(20) (empic.js:23-25)
    const h = f(spec.particle_charge * spec.dt / (2 * spec.particle_mass));
    const factor_r = 1 / spec.radius, factor_z = 1 / spec.height;
    const lit = function (x) { return f(Number(x.toFixed(20))); };
    const FR = fp64 ? factor_r : lit(factor_r), FZ = fp64 ? factor_z : lit(factor_z);
    const F_RZ = fp64 ? factor_r / factor_z : lit(factor_r / factor_z);
    const F_ZR = fp64 ? factor_z / factor_r : lit(factor_z / factor_r);
    const SF = f(spec.dt * 2.998e8);
    const C001 = f(0.001), C999 = f(0.999), C05 = f(0.5);

    const o = { n: n, nr: nr, nz: nz };
    const tex = function (cells) { return new Real(4 * cells); };
    o.pos_A = tex(n); o.vel_A = tex(n); o.rand_A = tex(n);
    o.pos_B = tex(n); o.vel_B = tex(n); o.rand_B = tex(n);
    o.entropy = tex(N_ENTROPY * N_ENTROPY);
    o.E = tex(nr * nz); o.B = tex(nr * nz); o.sink = tex(nr * nz); o.inv_cdf = tex(N_CDF * N_CDF);
    o.R1 = tex(nr * nz); o.R2 = tex(nr * nz); o.R3 = tex(nr * nz); o.A = tex(nr * nz);
    o.moments = tex(nr * nz); o.norm = tex(nr * nz); o.avg_A = tex(nr * nz); o.avg_B = tex(nr * nz);

    // 11x11 stamp (empic.js:949-971)
    o.stamp = new Float32Array(NSHAPE * NSHAPE);
    (function () {
        const mid = (NSHAPE - 1) / 2;
        let sum = 0;
        for (let j = 0; j < NSHAPE; j++) for (let i = 0; i < NSHAPE; i++) {
            const d = Math.sqrt(Math.pow(i - mid, 2) + Math.pow(j - mid, 2));
            o.stamp[i + NSHAPE * j] = Math.pow(Math.max(0.0, Math.cos(0.5 * Math.PI * d / mid)), 2);
            sum += o.stamp[i + NSHAPE * j];
        }
        for (let k = 0; k < NSHAPE * NSHAPE; k++) o.stamp[k] = o.stamp[k] / sum;
    })();

    // NEAREST + CLAMP_TO_EDGE (utilities.js:528-531); NaN selects texel 0
    function ngp(u, W) {
        const t = f(u * W);
        if (!(t >= 0)) return 0;
        if (t >= W) return W - 1;
        return Math.floor(t);
    }

    // ---- out.set (empic.js:1157-1350)
    o.set = function (value) {
        let i, j, p;
        if (value.E) for (i = 0; i < nr; i++) for (j = 0; j < nz; j++) {
            const c = 4 * (i + j * nr);
            o.E[c] = value.E[i][j][0]; o.E[c + 1] = value.E[i][j][1]; o.E[c + 2] = value.E[i][j][2]; o.E[c + 3] = 1.0;
        }
        if (value.B) for (i = 0; i < nr; i++) for (j = 0; j < nz; j++) {
            const c = 4 * (i + j * nr);
            o.B[c] = value.B[i][j][0]; o.B[c + 1] = value.B[i][j][1]; o.B[c + 2] = value.B[i][j][2]; o.B[c + 3] = 1.0;
        }
        if (value.position) for (p = 0; p < n; p++) {
            o.pos_A[4 * p] = value.position[p][0] * factor_r; o.pos_A[4 * p + 1] = value.position[p][1] * factor_r;
            o.pos_A[4 * p + 2] = value.position[p][2] * factor_z; o.pos_A[4 * p + 3] = 1.0;
        }
        if (value.position) o.pos_B.set(o.pos_A);
        if (value.velocity) for (p = 0; p < n; p++) {
            o.vel_A[4 * p] = value.velocity[p][0] * factor_r; o.vel_A[4 * p + 1] = value.velocity[p][1] * factor_r;
            o.vel_A[4 * p + 2] = value.velocity[p][2] * factor_z; o.vel_A[4 * p + 3] = 1.0;
        }
        if (value.velocity) o.vel_B.set(o.vel_A);
        if (value.sink_mask) for (i = 0; i < nr; i++) for (j = 0; j < nz; j++) o.sink[4 * (i + j * nr)] = value.sink_mask[i][j];
        if (value.source_pdf) buildInvCdf(value.source_pdf);
    };

    function buildInvCdf(pdf) {                                    // empic.js:1263-1339
        const lx = pdf.length, ly = pdf[0].length;
        const cdf_y = [], cdf_x = [];
        let sum_x = 0, i, j;
        for (i = 0; i < lx; i++) {
            cdf_y[i] = new Float64Array(ly);
            let sum_y = 0;
            for (j = 0; j < ly; j++) { sum_y += pdf[i][j]; cdf_y[i][j] = sum_y; }
            for (j = 0; j < ly; j++) cdf_y[i][j] /= sum_y;
            sum_x += sum_y; cdf_x[i] = sum_x;
        }
        for (i = 0; i < lx; i++) cdf_x[i] /= sum_x;
        const table = new Float32Array(4 * N_CDF * N_CDF);
        for (i = 0; i < N_CDF; i++) {
            const f1 = i / 511;
            let a = 0;
            while (a < lx && cdf_x[a] < f1) a++;
            const x = a === 0 ? (f1 / cdf_x[0]) / lx : (a === lx ? NaN : (a + (f1 - cdf_x[a - 1]) / (cdf_x[a] - cdf_x[a - 1])) / lx);
            if (x !== x) throw new TypeError('reference set({source_pdf}) throws for this pdf');
            const row = cdf_y[Math.min(lx - 1, Math.floor(x * lx))];
            for (j = 0; j < N_CDF; j++) {
                const f2 = j / 511;
                let b = 0;
                while (b < ly && row[b] < f2) b++;
                const y = b === 0 ? (f2 / row[0]) / ly : (b === ly ? NaN : (b + (f2 - row[b - 1]) / (row[b] - row[b - 1])) / ly);
                table[4 * (i + j * N_CDF)] = x; table[4 * (i + j * N_CDF) + 1] = y;
            }
        }
        for (i = 0; i < table.length; i++) o.inv_cdf[i] = table[i];
    }

    o.setRandomState = function (state) {
        if (state.entropy) for (let i = 0; i < o.entropy.length; i++) o.entropy[i] = state.entropy[i];
        if (state.rand) for (let i = 0; i < 4 * n; i++) o.rand_A[i] = state.rand[i];
    };

    o.addBZ = function (Bz) {                                       // empic.js:417-439, :1391
        const v = f(Bz);
        for (let c = 0; c < nr * nz; c++) { o.B[4 * c + 2] = f(o.B[4 * c + 2] + v); o.B[4 * c + 3] = f(o.B[4 * c + 3] + 1); }
    };

    // ---- out.precalc (empic.js:506-659, :1413-1434)
    o.precalc = function () {
        const B = o.B, E = o.E;
        for (let c = 0; c < nr * nz; c++) {
            const Bx = B[4 * c], By = B[4 * c + 1], Bz = B[4 * c + 2];
            const Ex = E[4 * c], Ey = E[4 * c + 1], Ez = E[4 * c + 2];
            const Bmag = f(Math.sqrt(f(f(f(Bx * Bx) + f(By * By)) + f(Bz * Bz))));
            const hB2 = f(f(f(h * h) * Bmag) * Bmag);
            const factor = f(2 / f(1 + hB2));
            const diag = f(1 - f(hB2 * factor));
            const fh = f(factor * h);
            o.R1[4 * c] = f(diag + f(f(f(fh * h) * Bx) * Bx));
            o.R1[4 * c + 1] = f(fh * f(Bz + f(f(h * Bx) * By)));
            o.R1[4 * c + 2] = f(f(fh * f(-By + f(f(h * Bx) * Bz))) * F_RZ);
            o.R1[4 * c + 3] = 1;
            o.R2[4 * c] = f(fh * f(-Bz + f(f(h * By) * Bx)));
            o.R2[4 * c + 1] = f(diag + f(f(f(fh * h) * By) * By));
            o.R2[4 * c + 2] = f(f(fh * f(Bx + f(f(h * By) * Bz))) * F_RZ);
            o.R2[4 * c + 3] = 1;
            o.R3[4 * c] = f(f(fh * f(By + f(f(h * Bz) * Bx))) * F_ZR);
            o.R3[4 * c + 1] = f(f(fh * f(-Bx + f(f(h * Bz) * By))) * F_ZR);
            o.R3[4 * c + 2] = f(diag + f(f(f(fh * h) * Bz) * Bz));
            o.R3[4 * c + 3] = 1;
            const a = f(h * f(2 - f(hB2 * factor)));
            const b = f(f(h * h) * factor);
            const cx = f(f(Ey * Bz) - f(Ez * By)), cy = f(f(Ez * Bx) - f(Ex * Bz)), cz = f(f(Ex * By) - f(Ey * Bx));
            const hd = f(h * f(f(f(Ex * Bx) + f(Ey * By)) + f(Ez * Bz)));
            const kx = opts.physical_a ? f(hd * Bx) : hd, ky = opts.physical_a ? f(hd * By) : hd, kz = opts.physical_a ? f(hd * Bz) : hd;
            const C = f(2.998e8);
            o.A[4 * c] = f(f(f(f(a * Ex) + f(b * f(cx + kx))) / C) * FR);
            o.A[4 * c + 1] = f(f(f(f(a * Ey) + f(b * f(cy + ky))) / C) * FR);
            o.A[4 * c + 2] = f(f(f(f(a * Ez) + f(b * f(cz + kz))) / C) * FZ);
            o.A[4 * c + 3] = 1;
        }
    };

    function stepRand(src, dst) {                                   // empic.js:783-820, K3
        const e = o.entropy;
        for (let p = 0; p < n; p++) {
            const q = 4 * p;
            let x0 = src[q + 2], x1 = src[q + 3];
            const s = 4 * (ngp(x0, N_ENTROPY) + N_ENTROPY * ngp(x1, N_ENTROPY));
            x0 = f(f(C999 * x0) + f(C001 * e[s + 2]));
            x1 = f(f(C999 * x1) + f(C001 * e[s + 3]));
            const m0 = f(src[q] + e[s]), m1 = f(src[q + 1] + e[s + 1]);
            dst[q] = m0 > 1 ? f(m0 - 1) : m0;
            dst[q + 1] = m1 > 1 ? f(m1 - 1) : m1;
            dst[q + 2] = f(f(4 * x0) * f(1 - x0));
            dst[q + 3] = f(f(4 * x1) * f(1 - x1));
        }
    }

    function stepVelocity(pos, vel, rnd, out) {                      // empic.js:729-778, K1
        const R1 = o.R1, R2 = o.R2, R3 = o.R3, A = o.A;
        for (let p = 0; p < n; p++) {
            const q = 4 * p;
            const x = pos[q], y = pos[q + 1];
            const r = f(Math.sqrt(f(f(x * x) + f(y * y))));
            const dx = f(x / r), dy = f(y / r);
            const vr = f(f(vel[q] * dx) + f(vel[q + 1] * dy));
            const va = f(f(vel[q + 1] * dx) - f(vel[q] * dy));
            const vz = vel[q + 2];
            const c = 4 * (ngp(r, nr) + nr * ngp(pos[q + 2], nz));
            const cx = f(f(f(f(R1[c] * vr) + f(R1[c + 1] * va)) + f(R1[c + 2] * vz)) + A[c]);
            const cy = f(f(f(f(R2[c] * vr) + f(R2[c + 1] * va)) + f(R2[c + 2] * vz)) + A[c + 1]);
            const cz = f(f(f(f(R3[c] * vr) + f(R3[c + 1] * va)) + f(R3[c + 2] * vz)) + A[c + 2]);
            if (pos[q + 3] > 0.5) {
                out[q] = f(f(cx * dx) - f(cy * dy)); out[q + 1] = f(f(cx * dy) + f(cy * dx)); out[q + 2] = cz; out[q + 3] = 1;
            } else {
                out[q] = f(C001 * f(f(2 * rnd[q]) - 1)); out[q + 1] = f(C001 * f(f(2 * rnd[q + 1]) - 1));
                out[q + 2] = f(C001 * f(f(2 * rnd[q + 2]) - 1)); out[q + 3] = C001;
            }
        }
    }

    function stepPosition(pos, vel, rnd, out) {                      // empic.js:692-726, K2
        const sink = o.sink, cdf = o.inv_cdf;
        for (let p = 0; p < n; p++) {
            const q = 4 * p;
            const nx = f(pos[q] + f(SF * vel[q])), ny = f(pos[q + 1] + f(SF * vel[q + 1])), nzp = f(pos[q + 2] + f(SF * vel[q + 2]));
            const r = f(Math.sqrt(f(f(nx * nx) + f(ny * ny))));
            const t = 4 * (ngp(rnd[q], N_CDF) + N_CDF * ngp(rnd[q + 1], N_CDF));
            const c = 4 * (ngp(r, nr) + nr * ngp(nzp, nz));
            if (sink[c] > 0.5) { out[q] = nx; out[q + 1] = ny; out[q + 2] = nzp; out[q + 3] = 1; }
            else { out[q] = cdf[t]; out[q + 1] = 0; out[q + 2] = cdf[t + 1]; out[q + 3] = 0; }
        }
    }

    o.step = function (ncalls) {                                     // empic.js:1436-1469
        for (let k = 0; k < (ncalls === undefined ? 1 : ncalls); k++) {
            stepRand(o.rand_A, o.rand_B);
            stepVelocity(o.pos_A, o.vel_A, o.rand_A, o.vel_B);
            stepPosition(o.pos_A, o.vel_B, o.rand_A, o.pos_B);
            stepRand(o.rand_B, o.rand_A);
            stepVelocity(o.pos_B, o.vel_B, o.rand_B, o.vel_A);
            stepPosition(o.pos_B, o.vel_A, o.rand_B, o.pos_A);
        }
    };

    // opts.raster_bits = b > 0: the point sprites as a rasteriser with b sub-pixel bits draws them (the C twin's deposit_raster,
    // pic_oracle_impl.h: clip coordinate 2u - 1, window position snapped to 2^-b pixel with ties to even and y running downwards,
    // left/top edges inclusive, cropped instead of discarded); first column / first row from the bottom, or null when dropped
    const rasterBits = opts.raster_bits | 0;
    function roundHalfEven(v) { const r = Math.round(v); return (r - v === 0.5 && r % 2 !== 0) ? r - 1 : r; }
    function rasterFirst(u, W, yDown) {
        const ndc = f(f(2 * u) - 1);
        const sc = 1 << rasterBits;
        const wb = f(f(W * 0.5) * sc), x0 = f(wb - f(sc * 0.5));
        const s = f(x0 + f(ndc * (yDown ? -wb : wb)));
        if (!(s > -1073741824 && s < 1073741824)) return null;
        const X = roundHalfEven(s), p0 = Math.ceil((X - 11 * sc / 2) / sc);
        return yDown ? W - 1 - (p0 + 10) : p0;
    }

    o.density = function () {                                        // empic.js:1471-1495
        const M = o.moments, pos = o.pos_A, vel = o.vel_A, w = o.stamp;
        M.fill(0);
        for (let p = 0; p < n; p++) {                                // K4 (empic.js:980-1035)
            const q = 4 * p;
            const x = pos[q], y = pos[q + 1], z = pos[q + 2];
            const r = f(Math.sqrt(f(f(x * x) + f(y * y))));
            let ic, jc;
            if (rasterBits) {
                const i0 = rasterFirst(r, nr, false), j0 = rasterFirst(z, nz, true);
                if (i0 === null || j0 === null || i0 >= nr || i0 + 10 < 0 || j0 >= nz || j0 + 10 < 0) continue;
                ic = i0 + 5; jc = j0 + 5;
            } else {
                if (!(r >= 0 && r <= 1 && z >= 0 && z <= 1)) continue;
                ic = Math.floor(f(r * nr)); jc = Math.floor(f(z * nz));
            }
            const dx = f(x / r), dy = f(y / r);
            const c0 = f(C001 * f(f(vel[q] * dx) + f(vel[q + 1] * dy)));
            const c1 = f(C001 * f(f(vel[q + 1] * dx) - f(vel[q] * dy)));
            const c2 = f(C001 * vel[q + 2]);
            const c3 = C001;
            for (let dj = -5; dj <= 5; dj++) {
                const j = jc + dj;
                if (j < 0 || j >= nz) continue;
                for (let di = -5; di <= 5; di++) {
                    const i = ic + di;
                    if (i < 0 || i >= nr) continue;
                    const wt = w[(di + 5) + NSHAPE * (5 - dj)];
                    const m = 4 * (i + nr * j);
                    M[m] = f(M[m] + f(c0 * wt)); M[m + 1] = f(M[m + 1] + f(c1 * wt));
                    M[m + 2] = f(M[m + 2] + f(c2 * wt)); M[m + 3] = f(M[m + 3] + f(c3 * wt));
                }
            }
        }
        const ratio = f(0.01), keep = f(1 - ratio);
        for (let j = 0; j < nz; j++) for (let i = 0; i < nr; i++) {  // K5, K6, K7 (empic.js:1042-1084, :1490)
            const c = 4 * (i + nr * j);
            const xc = f(f(i + C05) / nr);
            const a = M[c + 3];
            for (let k = 0; k < 4; k++) {
                const m = a > 0 ? (k < 3 ? f(M[c + k] / a) : a) : 0;
                o.norm[c + k] = f(f(f(1000 * m) * C05) / xc);
                o.avg_A[c + k] = f(f(ratio * o.norm[c + k]) + f(keep * o.avg_B[c + k]));
                o.avg_B[c + k] = o.avg_A[c + k];
            }
        }
    };
    return o;
}

module.exports = { makeOracle: makeOracle };

// ---- CLI: timing on a bounded sample (used by bench.py's cpu_baseline leg) and a
// dump mode used by the tests to compare against the C oracle.
if (require.main === module) {
    const args = process.argv.slice(2);
    const mode = args[0] || 'time';
    if (mode === 'time') {
        const side = Number(args[1] || 316), grid = Number(args[2] || 1024), seconds = Number(args[3] || 10);
        const spec = { radius: 1, height: 1, nr: grid, nz: grid, dt: 2e-9, nparticles: side, particle_mass: 1.67e-27, particle_charge: 1.602e-19 };
        const sim = makeOracle(spec);
        const n = sim.n;
        let s = 0x5EEDF051;
        const rnd = function () { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
        const pos = [], vel = [], sink = [], pdf = [];
        for (let p = 0; p < n; p++) {
            const rh = Math.max(Math.sqrt(rnd()), 1e-6), th = 2 * Math.PI * rnd();
            pos.push([rh * Math.cos(th), rh * Math.sin(th), rnd()]);
            vel.push([1e-3 * (rnd() + rnd() + rnd() - 1.5) * 2, 1e-3 * (rnd() + rnd() + rnd() - 1.5) * 2, 1e-3 * (rnd() + rnd() + rnd() - 1.5) * 2]);
        }
        for (let i = 0; i < grid; i++) {
            sink.push([]); pdf.push([]);
            for (let j = 0; j < grid; j++) { const v = (i === grid - 1 || j === 0 || j === grid - 1) ? 0 : 1; sink[i].push(v); pdf[i].push(v); }
        }
        sim.set({ position: pos, velocity: vel, sink_mask: sink, source_pdf: pdf });
        const ent = new Float32Array(4 * 1024 * 1024), rd = new Float32Array(4 * n);
        for (let i = 0; i < ent.length; i++) ent[i] = rnd();
        for (let i = 0; i < rd.length; i++) rd[i] = rnd();
        sim.setRandomState({ entropy: ent, rand: rd });
        sim.addBZ(0.01);
        let cycles = 0;
        const t0 = process.hrtime.bigint();
        let dt = 0;
        do { sim.precalc(); sim.step(); sim.density(); cycles++; dt = Number(process.hrtime.bigint() - t0) / 1e9; } while (dt < seconds && cycles < 1000);
        console.log(JSON.stringify({ value: 2 * n * cycles / dt, particles: n, grid: grid, cycles: cycles, seconds: dt, node: process.version }));
    }
}
