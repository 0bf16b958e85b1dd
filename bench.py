#!/usr/bin/env python3
"""bench.py — particle-updates/s of the push + deposit + solve cycle on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): axisymmetric (r,z) grid 1024 x 1024, 1e8 particles
per GPU, fp32, single species, synthetic uniform plasma: uniform in the
cylinder's volume, Maxwellian v_th = 1e-3 c, uniform Bz = 0.01 T, E = 0, sink frame,
uniform interior source, dt = 2e-9 s, proton m/q; entropy table and per-particle
random state injected from numpy's Philox generator, seed 0x5EEDF051.

One "step" = one frame of the reference's loop (fusionsim.js:170-178) plus the
field->coefficient stage: precalc() [solve stage, K8/K9], step() [two leap-frog
sub-steps = two particle-updates per particle, K3/K1/K2], density() [scatter K4,
normalise K5, EMA K6/K7].  value = 2 * N_particles_total * K / seconds.

Multi-GPU (reference-parity mode): particles are sharded by index range,
the grid tables are replicated, and the only exchange is one all-reduce (RCCL) of the
per-cell sums between the scatter and the stamp/normalise/EMA stage.  Per-GPU work is
fixed as N grows ("weak").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fusion-sim_amd"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0           # same guide: measured float4 copy
ALGO_BYTES_PER_UPDATE = 48      # push, SoA fp32: read 6 + write 6 scalars


def synthetic_inputs(n, spec, seed):
    """SURVEY.md 8(d) uniform plasma, float32 host arrays (physical units)."""
    rng = np.random.Generator(np.random.Philox(seed))
    pos = np.empty((n, 3), dtype=np.float32)
    vel = np.empty((n, 3), dtype=np.float32)
    chunk = 1 << 24
    for b in range(0, n, chunk):
        m = min(chunk, n - b)
        rh = np.maximum(np.sqrt(rng.random(m, dtype=np.float32)), np.float32(1e-6))
        th = np.float32(2 * np.pi) * rng.random(m, dtype=np.float32)
        pos[b:b + m, 0] = rh * np.cos(th) * np.float32(spec["radius"])
        pos[b:b + m, 1] = rh * np.sin(th) * np.float32(spec["radius"])
        pos[b:b + m, 2] = rng.random(m, dtype=np.float32) * np.float32(spec["height"])
        vel[b:b + m] = rng.standard_normal((m, 3), dtype=np.float32) * np.float32(1e-3)
    entropy = rng.random(1024 * 1024 * 4, dtype=np.float32)
    rand = rng.random((n, 4), dtype=np.float32)
    return pos, vel, entropy, rand


def scene_grids(nr, nz):
    sink = np.ones((nr, nz), dtype=np.float32)  # fusionsim.js:94-112
    sink[nr - 1, :] = 0
    sink[1:nr - 1, 0] = 0
    sink[1:nr - 1, nz - 1] = 0
    return sink, sink.copy()


def _time_oracle(spec, side, threads, seconds_target):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pic_oracle as po
    s = dict(spec, nparticles=side)
    n = side * side
    pos, vel, entropy, rand = synthetic_inputs(n, s, 0x5EEDF051)
    sink, pdf = scene_grids(s["nr"], s["nz"])
    sim = po.OracleSim(s, dtype=np.float32, threads=threads)
    sim.set(position=pos.astype(np.float64), velocity=vel.astype(np.float64), sink_mask=sink, source_pdf=pdf)
    sim.set_random_state(entropy, rand)
    sim.add_bz(0.01)
    cycles = 0
    t0 = time.perf_counter()
    while True:
        sim.precalc(); sim.step(); sim.density()
        cycles += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or cycles >= 200:
            break
    return 2.0 * n * cycles / dt, n, cycles, dt


def usable_cores(cap=64):
    """Threads the all-cores baseline may really use: the affinity mask, cut by the cgroup's CPU
    quota when there is one (a GPU box's container shares a large host), capped so the threaded
    deposit's private grids stay small."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def cpu_baseline(spec):
    """The build's CPU restatement of the reference shaders (oracle/, kind "port"; the reference has
    no CPU path) on bounded samples of the same workload: all host cores (OpenMP over particles,
    threaded deposit) as the headline baseline, one thread and the plain-JS twin beside it."""
    cores = usable_cores()
    grid = (spec["nr"], spec["nz"])
    v1, n1, c1, t1 = _time_oracle(spec, 1000, 1, 8.0)
    out = None
    if cores > 1:
        try:
            vm, nm, cm, tm = _time_oracle(spec, 2000, cores, 8.0)
            out = {"value": vm, "unit": "particle-updates/s", "cores": cores, "kind": "port",
                   "sample": "oracle/pic_oracle.c built with -fopenmp (CPU restatement of the reference's GLSL), fp32, %d threads, "
                             "grid %dx%d, %d particles, %d cycles of precalc+step+density in %.1f s"
                             % (cores, grid[0], grid[1], nm, cm, tm)}
        except Exception as e:  # the baseline is a report, never a reason to fail the bench
            out = None
            note = "OpenMP build unavailable: %s" % e
    single = {"value": v1, "unit": "particle-updates/s", "cores": 1, "kind": "port",
              "sample": "oracle/pic_oracle.c, fp32, 1 thread, grid %dx%d, %d particles, %d cycles in %.1f s"
                        % (grid[0], grid[1], n1, c1, t1)}
    if out is None:
        out = dict(single)
        if cores > 1:
            out["note"] = note
    out["single_thread"] = single
    out["js_twin"] = js_twin_baseline(spec["nr"])
    return out


def js_twin_baseline(grid, seconds_target=8.0):
    """The same restatement in plain JavaScript under node (oracle/pic_oracle.js): the
    closest analogue of a "JS engine on CPU" — the reference itself has none."""
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None:
        return {"value": None, "note": "node unavailable on this box"}
    try:
        raw = subprocess.check_output([node, os.path.join(ROOT, "oracle", "pic_oracle.js"), "time", "316", str(grid),
                                       str(seconds_target)], timeout=300)
        j = json.loads(raw.decode().strip().splitlines()[-1])
        return {"value": j["value"], "unit": "particle-updates/s", "cores": 1, "kind": "port",
                "sample": "oracle/pic_oracle.js under node %s, fp32 via Math.fround, grid %dx%d, %d particles, %d cycles in %.1f s"
                          % (j["node"], grid, grid, j["particles"], j["cycles"], j["seconds"])}
    except Exception as e:  # the baseline is a report, never a reason to fail the bench
        return {"value": None, "note": "node run failed: %s" % e}


def node_host_line(side, grid, steps, warmup, ctypes_ms_per_step):
    """The headline frame driven from the host north_star names — JavaScript: examples/bench_node.js under `node`, the scene
    uploaded from Float32Arrays through empic_native.js + the N-API addon, `precalc(); step(); density();` per frame timed in
    Node.  Run as a child process AFTER this process has given its particles back (both hold 1e8 particles on the card)."""
    import shutil
    import subprocess
    node = shutil.which("node")
    addon = os.path.join(ROOT, "fusion-sim_amd", "lib", "fusionpic_napi.node")
    if node is None or not os.path.exists(addon):
        return {"value": None, "note": "node %s, addon %s" % ("found" if node else "not on this machine", "built" if os.path.exists(addon) else "not built (make -C fusion-sim_amd napi)")}
    cmd = [node, "--max-old-space-size=8192", os.path.join(ROOT, "examples", "bench_node.js"), "--side", str(side), "--grid", str(grid), "--steps", str(steps), "--warmup", str(warmup)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        if p.returncode != 0:
            return {"value": None, "error": "exit %d: %s" % (p.returncode, p.stderr.decode()[-400:])}
        line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # a report, never a reason to lose the headline
        return {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
    line["what"] = ("the SAME frame as the headline, through the JavaScript host (Node N-API addon + empic_native.js) instead of Python ctypes; "
                    "vs_ctypes = its ms_per_step / the headline's")
    line["vs_ctypes"] = line["ms_per_step"] / ctypes_ms_per_step if ctypes_ms_per_step else None
    return line


def dense_sor_line(device, n_power=6, products=60):
    """SURVEY 8(f) next-4, the reference's dense iterative solver (matrix_webgl.js): products/s of
    x <- R x + C on an L = 16384 system (1 GiB iteration matrix, streamed once per product)."""
    import numpy as np
    from fusionpic import sor
    L = 4 * 4 ** n_power
    rng = np.random.default_rng(0x50F)
    A = ((rng.random((L, L), dtype=np.float32) - 0.5) * np.float32(1.0 / L))
    A[np.arange(L), np.arange(L)] = 1.0 + rng.random(L, dtype=np.float32)
    eq = sor.makeSORIterative({"n_power": n_power}, device=device)
    eq.set_matrix(A).set_b(rng.random(L, dtype=np.float32)).init_vector(np.zeros(L, dtype=np.float32)).prepare()
    eq.iterate(5)
    eq.sync()
    eq.resetStats()
    eq.profile(True)          # HIP events on the launch stream around the whole batch
    eq.iterate(products)
    eq.sync()
    st = eq.stats()
    per = st["seconds_iterate"] / st["iterations"]
    gbs = st["matrix_bytes"] / per / 1e9
    eq.close()
    return {"what": "dense weighted-Jacobi product x <- R x + C (matrix_webgl.js mv_product), n_power=%d, L=%d, float32, "
                    "bit-identical to the reference's pass order" % (n_power, L),
            "value": 1.0 / per, "unit": "products/s", "us_per_product": 1e6 * per,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": st["matrix_bytes"], "launches_timed": st["iterations"]}}


def es3d_scene(n, grid, seed=0x5EEDF051):
    """BASELINE configs[2] shape: periodic box of grid^3 nodes, warm electron plasma against a neutralising
    background: dx = Debye length, omega_p dt = 0.1, v_th = 1e-3 c, so v_th dt = 0.1 dx."""
    dt, vth = 1e-11, 1e-3
    c, eps0, me, qe = 2.998e8, 8.8541878128e-12, 9.109e-31, -1.602e-19
    dx = vth * c * dt / 0.1
    L = grid * dx
    wp = 0.1 / dt
    n0 = wp ** 2 * eps0 * me / qe ** 2
    spec = dict(radius=L, length_y=L, height=L, nr=grid, ny=grid, nz=grid, dt=dt, nparticles=0, count=n, particle_mass=me,
                particle_charge=qe, geometry="cart3d", solver="poisson_fft", macro_weight=n0 * L ** 3 / n)
    return spec, L, vth


def es3d_blocks(n, L, vth, seed=0x5EEDF051, block=1 << 24):
    """The population in blocks of <= 2^24 particles (a 5e8-particle host array would be 12 GB): positions are a
    Kronecker lattice frac(i * alpha) with the three R3 multipliers (uniform in the volume, and cheap: three integer
    multiplies per particle), velocities one Maxwellian block permuted and sign-flipped per block."""
    rng = np.random.Generator(np.random.Philox(seed))
    vplus = rng.standard_normal((min(block, n), 3), dtype=np.float32) * np.float32(vth)
    vminus = -vplus                                                           # pairs of blocks carry no net momentum
    mult = np.array([3518319155, 2882110345, 2360945575], dtype=np.uint32)   # round(2^32 / phi3^k), phi3 = 1.2207440846
    pos = np.empty((min(block, n), 3), dtype=np.float32)                      # one buffer, refilled per block
    idx = np.arange(min(block, n), dtype=np.uint32)
    tmp = np.empty(min(block, n), dtype=np.uint32)
    for b in range(0, n, block):
        m = min(block, n - b)
        for a in range(3):
            np.multiply(idx[:m] + np.uint32(b), mult[a], out=tmp[:m])
            np.right_shift(tmp[:m], np.uint32(8), out=tmp[:m])
            pos[:m, a] = tmp[:m]
            pos[:m, a] *= np.float32(L / 16777216.0)
        yield b, pos[:m], (vplus if (b // block) % 2 == 0 else vminus)[:m]


def es3d_line(device, n, grid, steps, warmup, stream=None, cpu=True):
    """Extension (no reference counterpart, parity unpinned): the self-consistent electrostatic cycle of BASELINE
    configs[2] — per sub-step one fused kernel (CIC gather, Boris, drift, int64 CIC deposit) and one Poisson solve
    (rocFFT around hand-written conversion, k-space and gradient kernels)."""
    import fusionpic as fp
    import torch
    spec, L, vth = es3d_scene(n, grid)
    sim = fp.makeCylindricalParticlePusher(spec, device=device)
    if stream is not None:
        sim.setStream(stream.cuda_stream)
    for first, pos, vel in es3d_blocks(n, L, vth):
        sim.setRange(first, position=pos, velocity=vel)
    sim.sort()
    sim.precalc()
    for _ in range(warmup):
        sim.step()
    sim.sync(); torch.cuda.synchronize()
    sim.resetStats(); sim.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    sim.sync(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = sim.stats()
    sim.destroy()
    substeps = 2 * steps
    push_ms = st["ms_push"] / max(1, st["step_launches"])
    solve_ms = st["ms_solve"] / max(1, st["solve_launches"])
    algo = 48.0 * n
    ach = algo / (push_ms * 1e-3) / 1e9 if push_ms > 0 else 0.0
    nodes = grid ** 3
    out = {
        "what": "spec.geometry='cart3d' (EXTENSION, parity unpinned: no reference counterpart): %d^3 periodic grid, %.1e electrons, "
                "fp32, CIC, Poisson solve every sub-step; bit-exact against the build's own oracle for a given field, solve within 2e-5"
                % (grid, n),
        "value": n * substeps / el, "unit": "particle-updates/s", "ms_per_substep": 1e3 * el / substeps,
        "kernel_ms_per_substep": {"push_gather_deposit": push_ms, "poisson_solve": solve_ms,
                                  "rebinning": st["ms_sort"] / substeps, "rebinning_passes": st["sort_passes"]},
        "roofline": {"bound": "hbm", "kernel": "push3_tiles_kernel<float> (gather + Boris + drift + deposit, one read and one write of the "
                                                "particle per sub-step)",
                     "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": algo, "avg_launch_ms": push_ms, "launches_timed": st["step_launches"]},
        "solve": {"algorithmic_bytes": 2 * 4 * nodes, "avg_ms": solve_ms,
                  "achieved_GBs": 2 * 4 * nodes / (solve_ms * 1e-3) / 1e9 if solve_ms > 0 else 0.0,
                  "note": "SURVEY 8(d): algorithmic = read rho + write phi; the five transform sweeps and the gradient move more (DESIGN 4.8)"},
        "cycle_bytes_per_update": 48.0 + 8.0 * nodes / n,
    }
    tr = measured_traffic({"workload": "c3", "particles": n, "grid": grid, "dtype": "f32"})
    if tr:
        out["roofline"]["traffic"] = tr.get("bytes_per_launch")
        out["roofline"]["traffic_source"] = "%s: %s" % (tr.get("file"), tr.get("source"))
        if push_ms > 0:
            out["roofline"]["traffic_rate_GBs"] = tr["bytes_per_launch"] / (push_ms * 1e-3) / 1e9
    else:
        out["roofline"]["traffic_source"] = "no committed PMC pass for this configuration (profiles/r*_c3_traffic.json)"
    if cpu:
        try:
            out["cpu_port"] = es3d_cpu_port()
        except Exception as e:  # a report, never a reason to fail the bench
            out["cpu_port"] = {"value": None, "note": str(e)}
    return out


def em_line(device, n, grid, steps, warmup, precision="fp64"):
    """Extension (no reference counterpart, parity unpinned): the full-EM cycle of BASELINE configs[4] on one GPU — node-centred
    gather + Boris + integer charge-conserving current deposit (both fields and the current accumulators of an 8x8x8-cell tile
    staged in LDS), Yee FDTD (B half, E, B half)."""
    import fusionpic as fp
    import torch
    c, eps0, me, qe, vth = 2.998e8, 8.8541878128e-12, 9.109e-31, -1.602e-19, 1e-3
    wp = 1e10
    dx = vth * c / wp                                   # Debye length
    L = grid * dx
    dt = 0.5 * dx / (c * 3 ** 0.5)
    n0 = wp ** 2 * eps0 * me / qe ** 2
    spec = dict(radius=L, length_y=L, height=L, nr=grid, ny=grid, nz=grid, dt=dt, nparticles=0, count=n, particle_mass=me, particle_charge=qe,
                geometry="cart3d", solver="yee", macro_weight=n0 * L ** 3 / n, precision=precision)
    sim = fp.makeCylindricalParticlePusher(spec, device=device)
    if n > 600_000_000:  # (minutes of numpy at this size: generated on the device, slab by slab)
        chunks = 16
        assert n % chunks == 0
        for r in range(chunks):
            pos, vel = c4_rank_particles(r, chunks, 0, n // chunks, L, vth, 1.0, device)
            sim.setRange(r * (n // chunks), position=pos, velocity=vel)
            del pos, vel
        torch.cuda.empty_cache()
    else:
        for first, pos, vel in es3d_blocks(n, L, vth):
            sim.setRange(first, position=pos, velocity=vel)
    sim.sort(); sim.precalc()
    for _ in range(warmup):
        sim.step()
    sim.sync(); torch.cuda.synchronize()
    sim.resetStats(); sim.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    sim.sync(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = sim.stats()
    device_bytes = st["bytes_particle_state"] + st["bytes_grid_state"]
    # density() of this mode (the charge grid is a diagnostic here, not part of the cycle): what a host that runs the reference's
    # frame — step() + density() — pays on top of the step; tiled over the EM tiles since round 4
    sim.density(); sim.sync(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        sim.density()
    sim.sync(); torch.cuda.synchronize()
    density_ms = 1e3 * (time.perf_counter() - t1) / 3
    sim.destroy()
    sub = 2 * steps
    esize = 8 if precision == "fp64" else 4
    push_ms = st["ms_push"] / max(1, st["step_launches"])
    algo = 12.0 * esize * n
    tr = measured_traffic({"workload": "em", "particles": n, "grid": grid, "dtype": "f64" if esize == 8 else "f32"})
    return {"what": "spec.solver='yee' (EXTENSION, parity unpinned): %d^3 periodic Yee lattice, %.1e electrons, %s"
                    % (grid, n, precision),
            "value": n * sub / el, "unit": "particle-updates/s", "ms_per_substep": 1e3 * el / sub,
            "kernel_ms_per_substep": {"push_gather_current": push_ms, "fdtd_b_e_b": st["ms_solve"] / max(1, st["solve_launches"]),
                                      "rebinning": st["ms_sort"] / sub},
            "roofline": {"bound": "hbm", "kernel": "em_push_tiles_kernel<%s>" % ("double" if esize == 8 else "float"), "achieved": algo / (push_ms * 1e-3) / 1e9 if push_ms else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / (push_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if push_ms else 0.0,
                         "traffic": tr.get("bytes_per_launch") if tr else None,
                         "traffic_source": ("%s: %s" % (tr.get("file"), tr.get("source"))) if tr else "no committed PMC pass for this configuration (profiles/r*_traffic.json)",
                         "algorithmic_bytes_per_launch": algo, "avg_launch_ms": push_ms},
            "density_ms": density_ms,
            "fdtd_algorithmic_bytes": 21 * esize * grid ** 3, "device_bytes": device_bytes}


def c4_scene(total, grid, world):
    """BASELINE configs[3] shape: periodic grid^3 box, electrons + protons (total/2 each) at the plasma parameters of
    es3d_scene; returns the spec of ONE handle holding `share` x world particles per species."""
    spec, L, vth = es3d_scene(total // 2, grid)
    return spec, L, vth, 1836.15267 * spec["particle_mass"], -spec["particle_charge"]


def c4_rank_particles(rank, world, species, share, L, vth, mass_ratio, device):
    """Particles [rank*share, (rank+1)*share) of a species, generated on the device (2e9 particles would be minutes of
    numpy): the ones that start in z-slab `rank` of `world`.  x, y on the Kronecker lattice of es3d_blocks,
    z = (rank + frac) * L / world with a 20-bit frac (exact in float32, so a particle never rounds into the next
    slab); velocities a Maxwellian block repeated with alternating sign."""
    import torch
    dev = torch.device("cuda", device)
    ids = torch.arange(rank * share, (rank + 1) * share, dtype=torch.int64, device=dev)
    pos = torch.empty((share, 3), dtype=torch.float32, device=dev)
    for a, mult in enumerate((3518319155, 2882110345, 2360945575)):
        t = (ids * mult + 0x9E3779B9 * (species + 1)) & 0xFFFFFFFF           # frac(i * alpha) in 32 bits
        if a < 2:
            pos[:, a] = (t >> 8).to(torch.float32) * (L / 16777216.0)
        else:
            pos[:, a] = ((t >> 12).to(torch.float32) * (1.0 / 1048576.0) + float(rank)) * (L / world)
        del t
    del ids
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC4 + species)
    block = min(share, 1 << 24)
    half = torch.randn((block, 3), dtype=torch.float32, device=dev, generator=gen) * (vth / mass_ratio ** 0.5)
    pair = torch.cat([half, -half])
    reps = (share + pair.shape[0] - 1) // pair.shape[0]
    vel = pair.repeat(reps, 1)[:share].contiguous()
    torch.cuda.synchronize(dev)
    return pos, vel


def c4_line(device, total, grid, world, steps, warmup, precision="fp32", skip_single=False, ghost=4, solver="poisson_fft"):
    """Extension, parity unpinned: BASELINE configs[3] (two species, grid^3, `total` particles, z-slab decomposition over
    `world` ranks) exercised at full size on ONE GPU: (1) one handle holding everything = the single-GPU strong-scaling
    baseline; (2) the `world` ranks of the decomposition as handles of this process (fpic_group_*: the exchange is
    device-to-device copies of the same buffers RCCL sends), which gives each rank's kernel time at its real share
    and the bytes every exchange moves.  What one GPU cannot give is the link time of (2): it is priced in DESIGN.md 6
    from these byte counts."""
    import fusionpic as fp
    import torch
    spec, L, vth, mi, qi = c4_scene(total, grid, world)
    em = solver == "yee"
    solve_mode = os.environ.get("FPIC_BENCH_SOLVE", "interface")   # "interface": fes_tri.hpp (no transposition); "1": the transposed spectrum
    solve_mode = "interface" if solve_mode == "interface" and not em else True
    nspecies = 1 if em else 2          # configs[4] names no second species: electrons against a neutralising background
    share = total // nspecies // world
    # a thermal electron of this scene moves 0.08 cells per sub-step: with a migration every 2 G sub-steps the 6.3-sigma tail of
    # 2e9 particles outran G = 4 ghost planes (5 or 6 particles lost per run in rounds 2 and 3); 2 G - 2 leaves 8.4 sigma
    migrate_every = max(2, 2 * ghost - 2)
    if em:  # the time step of the Yee lattice (c dt = dx / 2 sqrt 3): a thermal electron moves 3e-4 cells per sub-step
        dx = L / grid
        spec = dict(spec, solver="yee", dt=0.5 * dx / (2.998e8 * 3 ** 0.5), macro_weight=spec["macro_weight"] * 2)
        migrate_every = 64
    def particles(r, sp):
        return c4_rank_particles(r, world, sp, share, L, vth, 1.0 if sp == 0 else mi / spec["particle_mass"], device)

    def run(step_fn, sync_fn, reset_fn):
        for _ in range(warmup):
            step_fn()
        sync_fn(); torch.cuda.synchronize()
        reset_fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync_fn(); torch.cuda.synchronize()
        return time.perf_counter() - t0

    out = {"what": ("BASELINE configs[4] shape on ONE GPU (EXTENSION, parity unpinned): %d^3 Yee lattice, %.1e electrons, %s, full EM"
                    % (grid, share * world, precision)) if em else
                   ("BASELINE configs[3] shape on ONE GPU (EXTENSION, parity unpinned): %d^3 periodic grid, %.1e electrons + %.1e protons, %s, "
                    "Poisson solve every sub-step" % (grid, share * world, share * world, precision)),
           "unit": "particle-updates/s"}
    n_all = nspecies * share * world
    sub = 2 * steps
    if not skip_single:
        one = fp.makeCylindricalParticlePusher(dict(spec, count=share * world), device=device, precision=precision)
        if nspecies == 2:
            one.addSpecies(mi, qi, share * world)
        for sp in range(nspecies):
            for r in range(world):
                p, v = particles(r, sp)
                one.setRange(r * share, position=p, velocity=v, species=sp)
                del p, v
        torch.cuda.empty_cache()
        one.sort(); one.precalc()
        def reset_one():
            one.resetStats(); one.profile(True)
        el = run(one.step, one.sync, reset_one)
        st = one.stats()
        fixed_sum = None
        out["single_handle"] = {"value": n_all * sub / el, "ms_per_substep": 1e3 * el / sub,
                                "kernel_ms_per_substep": {"push_all_species": st["ms_push"] / sub, "field_solve": st["ms_solve"] / max(1, st["solve_launches"]),
                                                          "rebinning": st["ms_sort"] / sub},
                                "device_bytes": st["bytes_particle_state"] + st["bytes_grid_state"]}
        one.destroy()
        del one
    ranks = []
    cap = int(share * 1.25)
    for r in range(world):
        s_ = fp.makeCylindricalParticlePusher(dict(spec, count=cap), device=device, precision=precision)
        if nspecies == 2:
            s_.addSpecies(mi, qi, cap)
        s_.domainInit(r, world, ghost_planes=ghost, migrate_every=migrate_every, distributed_solve=solve_mode)
        for sp in range(nspecies):
            p, v = particles(r, sp)
            s_.domainSet(p, v, first_id=r * share, species=sp)
            del p, v
        ranks.append(s_)
    torch.cuda.empty_cache()
    group = fp.BoxGroup(ranks)
    group.precalc()
    def sync_all():
        for s_ in ranks:
            s_.sync()

    def reset_all():
        for s_ in ranks:
            s_.resetStats(); s_.profile(True)
    mig0 = [0]

    def reset_and_note():
        reset_all()
        mig0[0] = sum(s_.domainStats()["migrated"] for s_ in ranks)
    el = run(group.step, sync_all, reset_and_note)
    sts = [s_.stats() for s_ in ranks]
    dom = [s_.domainStats() for s_ in ranks]
    esz = 4 if precision == "fp32" else 8
    plane = grid * grid
    out["decomposed_in_process"] = {
        "ranks": world, "serial_ms_per_substep_all_ranks": 1e3 * el / sub,
        "note": "the ranks share one GPU and run one after the other: per-rank kernel times are in the rocprofv3 kernel stats of this command "
                "(profiles/r03_c4_kernel_stats.csv: the launches of the decomposed part / ranks), not in this wall-clock figure",
        "particles_migrated_per_substep": (sum(d["migrated"] for d in dom) - mig0[0]) / float(sub), "lost": sum(d["lost"] for d in dom),
        "ghost_planes": ghost, "migrate_every": migrate_every,
        "exchange_bytes_per_rank_per_substep":
            {"current_ghost_planes_int64_reduce": 2 * (ghost + 2) * plane * 3 * 8,
             # (round 4, chained lattice step: a rank forms the half-time B of its halo planes itself; E goes one plane deeper into the upper halo)
             "lattice_halo_copies_E": (2 * (ghost + 2) + 1) * plane * 4 * esz} if em else
            {"ghost_planes_int64_reduce": (2 * ghost + 1) * plane * 8,
             # (rows of the half spectrum are padded to whole 128-byte tiles in the library's own transform buffers)
             # (one formula with box_workload; ADVICE r03: the two used different leading factors)
             **({"interface_planes_all_gather": (2 * (-(-(grid // 2 + 1) // (64 // esz)) * (64 // esz)) * grid + grid // world) * 2 * esz * (world - 1)}
                if solve_mode == "interface" else
                {"fft_transposes_all_to_all": 2 * (-(-(grid // 2 + 1) // (64 // esz)) * (64 // esz)) * grid * (grid // world) * 2 * esz * (world - 1) // world}),
             "potential_planes": (2 * ghost + 3) * plane * esz},
        "decomposed_solve": "interface system along z (fes_tri.hpp)" if solve_mode == "interface" else "transposed spectrum",
    }
    for s_ in ranks:
        s_.destroy()
    return out


def box_workload(args, rank, world, local_rank, dist, steps=None, warmup=None, cpu=True):
    """BASELINE configs[3] (or, with --c4-solver yee, configs[4]'s shape) as ONE decomposed run over the launcher's
    ranks — this process is rank `rank` of `world` z-slabs, the exchange (ghost planes, FFT transposes or lattice halos,
    migrating particles) runs inside libfusionpic.so over its own RCCL communicator.  A fixed total population:
    `"scaling": "strong"`.  (World of one = one handle holding everything.)  Returns the block on rank 0, None elsewhere.
    There is no other transport: if the library's communicator cannot be set up this raises on every rank."""
    import fusionpic as fp
    import torch
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    total, grid, ghost = int(args.c4_particles), args.c4_grid, args.c4_ghost
    em = args.c4_solver == "yee"
    solve_mode = os.environ.get("FPIC_BENCH_SOLVE", "interface")   # "interface": fes_tri.hpp (no transposition); "1": the transposed spectrum
    solve_mode = "interface" if solve_mode == "interface" and not em and world <= 8 else True
    spec, L, vth, mi, qi = c4_scene(total, grid, world)
    nspecies = 1 if em else 2
    share = total // nspecies // world
    # a thermal electron of this scene moves 0.08 cells per sub-step: with a migration every 2 G sub-steps the 6.3-sigma tail of
    # 2e9 particles outran G = 4 ghost planes (5 or 6 particles lost per run in rounds 2 and 3); 2 G - 2 leaves 8.4 sigma
    migrate_every = max(2, 2 * ghost - 2)
    if em:
        spec = dict(spec, solver="yee", dt=0.5 * (L / grid) / (2.998e8 * 3 ** 0.5), macro_weight=spec["macro_weight"] * 2)
        migrate_every = 64
    cap = share if world == 1 else int(share * 1.25)
    sim = fp.makeCylindricalParticlePusher(dict(spec, count=cap), device=local_rank, precision=args.c4_precision)
    if nspecies == 2:
        sim.addSpecies(mi, qi, cap)
    comm = {"transport": "none (one process, one handle)", "rank": 0, "world": 1}
    if dist is not None:
        box = [None]
        if rank == 0:
            try:
                box[0] = fp.commUniqueId()
            except Exception as e:
                box[0] = "error: %s" % e
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], bytes):
            raise RuntimeError("the library's RCCL communicator cannot be set up (%s); the decomposed run has no other transport" % box[0])
        sim.commInit(box[0], rank, world)
        r_, w_ = sim.commInfo()   # what the library itself reports (fpic_comm_info)
        comm = {"transport": "libfusionpic.so's own RCCL communicator (fpic_comm_init)", "rank": r_, "world": w_}
        if w_ != world:
            raise RuntimeError("fpic_comm_info reports a world of %d, the launcher has %d ranks" % (w_, world))
    sim.domainInit(rank, world, ghost_planes=ghost, migrate_every=migrate_every, distributed_solve=solve_mode if world > 1 else False)
    for sp in range(nspecies):
        p, v = c4_rank_particles(rank, world, sp, share, L, vth, 1.0 if sp == 0 else mi / spec["particle_mass"], local_rank)
        sim.domainSet(p, v, first_id=rank * share, species=sp)
        del p, v
    torch.cuda.empty_cache()

    def fence():
        sim.sync(); torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(); torch.cuda.synchronize()
    sim.precalc()
    for _ in range(warmup):
        sim.step()
    fence()
    sim.resetStats(); sim.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    fence()
    elapsed = time.perf_counter() - t0
    st, dom = sim.stats(), sim.domainStats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=getattr(args, "reduce_device", "cuda"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sim.destroy()
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    sub = 2 * steps
    n_all = nspecies * share * world
    esz = 4 if args.c4_precision == "fp32" else 8
    push_ms = st["ms_push"] / sub
    algo = 12.0 * esz * nspecies * share     # one read and one write of the six coordinates per particle and sub-step
    plane = grid * grid
    if world == 1:
        exchanges = {}
    elif em:
        exchanges = {"current_ghost_planes_int64_reduce": 2 * (ghost + 2) * plane * 3 * 8, "lattice_halo_copies_E_and_B": 2 * 2 * (ghost + 2) * plane * 4 * esz}
    else:
        exchanges = {"ghost_planes_int64_reduce": (2 * ghost + 1) * plane * 8,
                     # (rows of the half spectrum are padded to whole 128-byte tiles in the library's own transform buffers)
                     # the decomposed direction of the solve: two transpositions of the rank's half spectrum (sent bytes), or — the
                     # interface solve, csrc/fes_tri.hpp — one all-gather of two planes of it per rank (received bytes)
                     **({"interface_planes_all_gather": (2 * (-(-(grid // 2 + 1) // (64 // esz)) * (64 // esz)) * grid + grid // world) * 2 * esz * (world - 1)}
                        if solve_mode == "interface" else
                        {"fft_transposes_all_to_all": 2 * (-(-(grid // 2 + 1) // (64 // esz)) * (64 // esz)) * grid * (grid // world) * 2 * esz * (world - 1) // world}),
                     "potential_planes": (2 * ghost + 3) * plane * esz,
                     "migration_records_32B": 32 * dom["migrated"] // max(1, sub)}
    out = {
        "metric": "particle-updates/sec (push+deposit+solve)", "value": n_all * sub / elapsed, "unit": "particle-updates/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32" if esz == 4 else "f64", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[4] shape: %d^3 Yee lattice, %.1e electrons in total, full EM" if em else
                                "BASELINE configs[3]: %d^3 periodic grid, %.1e particles in total (electrons + protons), Poisson solve every sub-step")
                               % (grid, n_all) + "; EXTENSION, parity unpinned (no reference counterpart); one step = 2 sub-steps",
                   "parallelism": "z-slab decomposition x%d inside libfusionpic.so over RCCL (ghost planes %d, migration every %d sub-steps%s)"
                                  % (world, ghost, migrate_every, "" if em or world == 1 else (", decomposed solve: interface system along z, one all-gather of two planes per rank" if solve_mode == "interface" else ", slab-decomposed FFT with two transpositions"))},
        "comm": comm,
        "roofline": {"bound": "hbm", "kernel": "em_push_tiles_kernel" if em else "push3_tiles_kernel", "achieved": algo / (push_ms * 1e-3) / 1e9 if push_ms else 0.0,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / (push_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if push_ms else 0.0, "traffic": None,
                     "algorithmic_bytes_per_launch": algo, "avg_launch_ms": push_ms, "note": "rank 0's push of all its species per sub-step"},
        "kernel_ms_per_substep_rank0": {"push": push_ms, "field_solve": st["ms_solve"] / sub, "rebinning_and_migration": st["ms_sort"] / sub},
        "exchange_bytes_per_rank_per_substep": exchanges,
        "migrated_rank0": dom["migrated"], "lost_rank0": dom["lost"], "device_bytes_rank0": st["bytes_particle_state"] + st["bytes_grid_state"],
        "cpu_baseline": None,
    }
    tr = measured_traffic({"workload": "c4", "particles_per_gpu": nspecies * share, "grid": grid, "dtype": out["dtype"], "solver": args.c4_solver})
    if tr:
        out["roofline"]["traffic"] = tr.get("bytes_per_launch")
        out["roofline"]["traffic_source"] = "%s: %s" % (tr.get("file"), tr.get("source"))
    else:
        out["roofline"]["traffic_source"] = "no committed PMC pass for this configuration (profiles/r*_traffic.json)"
    if cpu and not em:
        try:  # the build's CPU oracle of the same cycle on a bounded sample (kind "port": the reference has no such mode)
            out["cpu_baseline"] = es3d_cpu_port()
        except Exception as e:  # a report, never a reason to fail the bench
            out["cpu_baseline"] = {"value": None, "note": str(e)}
    return out


def es3d_cpu_port(seconds_target=5.0):
    """The build's own CPU oracle of the same cycle (oracle/es3d_oracle.c, OpenMP) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import es3d_oracle as eo
    cores = usable_cores()
    n, grid = 2_000_000, 64
    spec, L, vth = es3d_scene(n, grid)
    sim = eo.OracleES3D(spec, np.float32, threads=cores)
    _, pos, vel = next(es3d_blocks(n, L, vth))
    sim.set(position=pos.astype(np.float64), velocity=vel.astype(np.float64))
    sim.precalc()
    k, t0 = 0, time.perf_counter()
    while True:
        sim.substep(); k += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or k >= 100:
            break
    return {"value": n * k / dt, "unit": "particle-updates/s", "cores": cores, "kind": "port",
            "sample": "oracle/es3d_oracle.c with -fopenmp, fp32, %d threads, %d^3 grid, %d particles, %d sub-steps in %.1f s" % (cores, grid, n, k, dt)}


def measured_traffic(config):
    """HBM bytes per push launch from the committed PMC passes (scripts/pmc_bench.sh,
    profiles/*_traffic.json): rocprofv3 cannot run inside the timed process, so the figure is the
    builder-side measurement of the SAME configuration (particles per GPU, grid, generator,
    precision) — for any other configuration there is none and `traffic` stays null."""
    best = None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("config") == config:
            best = dict(j, file=os.path.relpath(f, ROOT))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--side", type=int, default=10000, help="particle texture side per GPU (count = side^2)")
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rng", choices=["reference", "counter"], default="reference",
                    help="reference = the reference's entropy-table generator (the drop-in, the headline); "
                         "counter = the Philox extension mode (SURVEY.md 8(d))")
    ap.add_argument("--no-overlap", action="store_true", help="multi-GPU: all-reduce on the pusher's stream, no overlap")
    ap.add_argument("--total-particles", type=float, default=0,
                    help="strong scaling: this many particles in total, split over the ranks by contiguous index ranges "
                         "(fusionpic.multi.shard_bounds); default 0 = weak scaling with --side^2 particles per GPU")
    ap.add_argument("--comm", choices=["lib", "torch"], default="lib",
                    help="multi-GPU exchange: lib = the library's own RCCL communicator (fpic_comm_*, what a JavaScript host uses); "
                         "torch = torch.distributed all-reduce on the buffer's device address (fusionpic.multi.ShardedPusher)")
    ap.add_argument("--c3-particles", type=int, default=500_000_000, help="extensions.c3: particles of the electrostatic box (BASELINE configs[2]: 5e8)")
    ap.add_argument("--c3-grid", type=int, default=256, help="extensions.c3: nodes per axis (BASELINE configs[2]: 256)")
    ap.add_argument("--only-c3", action="store_true", help="development: measure extensions.c3 alone and print it")
    ap.add_argument("--only-em", action="store_true", help="development: measure the full-EM extension alone (--c3-particles, --c3-grid, --em-precision)")
    ap.add_argument("--em-precision", choices=["fp32", "fp64"], default="fp64")
    ap.add_argument("--workload", choices=["rz", "box"], default="rz",
                    help="rz = the headline (reference-parity pusher, particle shards); box = BASELINE configs[3]/[4] as one z-slab-decomposed "
                         "run over the ranks (strong scaling; --c4-particles, --c4-grid, --c4-solver, --c4-precision, --c4-ghost)")
    ap.add_argument("--only-c4", action="store_true", help="development: BASELINE configs[3] shape on one GPU (one handle, then the in-process decomposition)")
    ap.add_argument("--c4-particles", type=float, default=2e9, help="--only-c4: total particles (two species, half each)")
    ap.add_argument("--c4-grid", type=int, default=512)
    ap.add_argument("--c4-ranks", type=int, default=8)
    ap.add_argument("--c4-skip-single", action="store_true")
    ap.add_argument("--c4-solver", choices=["poisson_fft", "yee"], default="poisson_fft", help="--only-c4: 'yee' rehearses the full-EM decomposition (configs[4])")
    ap.add_argument("--c4-precision", choices=["fp32", "fp64"], default="fp32")
    ap.add_argument("--c4-ghost", type=int, default=4)
    ap.add_argument("--no-sink", action="store_true", help="development: no sink cells, hence no re-injected particles (ablation of the re-binning trigger)")
    ap.add_argument("--sort-interval", type=int, default=0, help="development: fixed re-binning period in frames (0 = the adaptive trigger)")
    ap.add_argument("--no-extensions", action="store_true", help="skip the extension measurements at N=1 (counter RNG, dense solver)")
    ap.add_argument("--no-c5", action="store_true", help="skip extensions.c5_one_gpu (BASELINE configs[4] at full size on one GPU: about a minute)")
    ap.add_argument("--no-strong-c4", action="store_true", help="development: skip the strong_c4 block (BASELINE configs[3] as one decomposed run over the ranks)")
    ap.add_argument("--bootstrap", choices=["nccl", "gloo"], default="nccl",
                    help="the launcher-side process group that hands the library's unique id round and takes the maximum of the ranks' clocks. "
                         "nccl (= RCCL, the default: one GPU per rank).  gloo is for REHEARSAL ONLY: every rank runs on the visible device "
                         "LOCAL_RANK %% device_count, so that N processes can share one GPU with a stand-in bound through FPIC_RCCL_LIBRARY "
                         "(tests/fake_rccl/fake_rccl_shm.cpp) and run the exact process topology of the N-GPU launch; its line says so and is no measurement")
    args = ap.parse_args()

    import torch
    import fusionpic as fp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run the collective path is taken even for a world of one, so a
    # 1-GPU box can rehearse exactly what the N-GPU launch executes
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    if args.gpus != world:
        if distributed or args.gpus != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, world, args.gpus))
    rehearsal = args.bootstrap == "gloo"
    if rehearsal:  # (device_count() does not initialise the GPU; the launcher started this process before any GPU call)
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if distributed and rehearsal:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
    elif distributed:
        import torch.distributed as dist
        # the exchange overlaps the next frame's push (ShardedPusher): give RCCL's stream priority so its
        # few workgroups are placed as soon as a push workgroup retires
        opts = None
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except (AttributeError, TypeError):
            pass
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
    # the launcher-side reductions (a flag, a clock): device tensors over RCCL, host tensors over gloo
    args.reduce_device = "cpu" if rehearsal else "cuda"
    rehearsal_note = None
    if rehearsal:
        rehearsal_note = ("REHEARSAL, not a measurement: --bootstrap gloo, %d processes on visible device(s) %s, the library's communicator bound to "
                          "FPIC_RCCL_LIBRARY=%s" % (world, "0..%d" % (torch.cuda.device_count() - 1) if torch.cuda.device_count() > 1 else "0",
                                                    os.environ.get("FPIC_RCCL_LIBRARY", "(unset: the real RCCL)")))

    if args.workload == "box":
        line = box_workload(args, rank, world, local_rank, dist if distributed else None, cpu=not args.no_cpu_baseline)
        if rank == 0:
            if rehearsal_note:
                line["rehearsal"] = rehearsal_note
            print(json.dumps(line), flush=True)
        if distributed:
            dist.destroy_process_group()
        return

    if args.only_c3:
        print(json.dumps({"c3": es3d_line(local_rank, args.c3_particles, args.c3_grid, args.steps, args.warmup, cpu=not args.no_cpu_baseline)}), flush=True)
        return

    if args.only_c4:
        print(json.dumps({"c4": c4_line(local_rank, int(args.c4_particles), args.c4_grid, args.c4_ranks, args.steps, args.warmup,
                                        precision=args.c4_precision, skip_single=args.c4_skip_single, ghost=args.c4_ghost, solver=args.c4_solver)}), flush=True)
        return

    if args.only_em:
        print(json.dumps({"em": em_line(local_rank, args.c3_particles, args.c3_grid, args.steps, args.warmup, args.em_precision)}), flush=True)
        return

    spec = dict(radius=1.0, height=1.0, nr=args.grid, nz=args.grid, dt=2e-9, nparticles=args.side,
                particle_mass=1.67e-27, particle_charge=1.602e-19)
    n_local = args.side * args.side
    strong = args.total_particles > 0
    if strong:  # a fixed population, sharded by contiguous index ranges
        from fusionpic.multi import shard_bounds
        lo, hi = shard_bounds(int(args.total_particles), rank, world)
        n_local = hi - lo
        spec["nparticles"] = 1
    # every rank draws its own shard of the global particle population
    pos, vel, entropy, rand = synthetic_inputs(n_local, spec, 0x5EEDF051 + 7919 * rank)
    if distributed:  # replicated tables must be identical on every rank
        _, _, entropy, _ = synthetic_inputs(1, spec, 0x5EEDF051)
    sink, pdf = scene_grids(spec["nr"], spec["nz"])
    if args.no_sink:
        sink = np.ones_like(sink)

    stream = torch.cuda.Stream(device=local_rank)

    def build(rng_mode, shape="ref11"):
        # counter mode: no gather hides the LDS atomics of the fused sums, so only the tile census and
        # the re-binning stay in the push there (spec.unfused_deposit = 2)
        s_ = fp.makeCylindricalParticlePusher(spec, device=local_rank, rng=rng_mode, seed=0x5EEDF051, count=n_local if strong else 0, sort_interval=args.sort_interval,
                                              fuse_deposit="census" if (rng_mode == "counter" or shape == "cic") else True, shape=shape)
        s_.setStream(stream.cuda_stream)
        s_.set(position=pos, velocity=vel, sink_mask=sink, source_pdf=pdf)
        if rng_mode == "reference":
            s_.setRandomState(entropy, rand)
        s_.addBZ(0.01)
        return s_

    sim = build(args.rng)
    sim.precalc()
    sim.sync()
    t_bin = time.perf_counter()
    sim.sort()  # first binning of the randomly ordered upload belongs to setup, like the upload itself
    sim.sync()
    first_binning_ms = 1e3 * (time.perf_counter() - t_bin)

    sharded = None
    comm_fallback = None
    comm = {"transport": "none (one process, one handle)", "rank": 0, "world": 1}
    if distributed and args.comm == "lib":
        # the library's own communicator: rank 0 draws the RCCL id, the host hands it round (here over the
        # launcher's process group; a JavaScript host uses a file or its own channel), and density() then
        # all-reduces the per-cell sums inside libfusionpic.so on a side stream, overlapped with the next push
        comm_note = None
        box = [None]
        if rank == 0:
            try:
                box[0] = fp.commUniqueId()
            except Exception as e:  # e.g. no RCCL to bind
                box[0] = "error: %s" % e
        dist.broadcast_object_list(box, src=0)
        ok = 0
        if isinstance(box[0], bytes):
            try:
                sim.commInit(box[0], rank, world, overlap=not args.no_overlap)
                ok = 1
            except Exception as e:
                comm_note = str(e)
        else:
            comm_note = box[0]
        flag = torch.tensor([ok], dtype=torch.int32, device=args.reduce_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # NEVER silently: a value measured over torch.distributed is not the product's transport.  With more than
            # one GPU the run ends here, non-zero, with a line that says so; a world of one (a rehearsal on a one-GPU
            # box) may go on through torch.distributed, and the line carries "comm_fallback"
            comm_fallback = "the library's RCCL communicator could not be set up: %s" % (comm_note or "on another rank")
            if ok:
                sim.commDestroy()
            if world > 1:
                if rank == 0:
                    print(json.dumps({"metric": "particle-updates/sec (push+deposit+solve)", "value": None, "unit": "particle-updates/s", "n_gpus": world,
                                      "comm_fallback": comm_fallback, "error": "no measurement: a multi-GPU value must come from libfusionpic.so's own communicator"}),
                          flush=True)
                sim.destroy()
                dist.destroy_process_group()
                raise SystemExit(3)
            args.comm = "torch"
        else:
            r_, w_ = sim.commInfo()   # what the library itself reports (fpic_comm_info)
            comm = {"transport": "libfusionpic.so's own RCCL communicator (fpic_comm_init)", "rank": r_, "world": w_}
            if w_ != world:
                raise SystemExit("fpic_comm_info reports a world of %d, the launcher has %d ranks" % (w_, world))
    elif distributed:
        comm_fallback = "--comm torch was asked for on the command line (development): torch.distributed all-reduce, not the product's transport"
    if distributed and args.comm != "lib":
        from fusionpic.multi import ShardedPusher, device_tensor_view
        ptr, nbytes = sim.deviceBuffer()
        sums = device_tensor_view(ptr, nbytes, torch.device("cuda", local_rank))
        sharded = ShardedPusher(sim, sums, stream=stream, overlap=not args.no_overlap)
        comm = {"transport": "torch.distributed (NOT the product's transport)", "rank": rank, "world": world}

    def cycle():
        sim.precalc()
        sim.step()
        if sharded is not None:
            sharded.density()
        else:
            sim.density()

    def fence():
        sim.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        cycle()
    fence()
    sim.resetStats()
    sim.profile(True)  # HIP events on the launch stream, read after the timed region
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle()
    fence()
    elapsed = time.perf_counter() - t0
    st = sim.stats()
    sim.profile(False)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_total = int(args.total_particles) if strong else n_local * world
        updates = 2.0 * n_total * args.steps
        push_ms = st["ms_push"] / max(1, st["step_launches"])
        achieved = ALGO_BYTES_PER_UPDATE * 2.0 * n_local / (push_ms * 1e-3) / 1e9 if push_ms > 0 else 0.0
        out = {
            "metric": "particle-updates/sec (push+deposit+solve)",
            "value": updates / elapsed,
            "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "2D axisymmetric (r,z) %dx%d grid, %.0e particles per GPU, single species, reference stamp "
                            "(11x11) deposit, uniform Bz=0.01 T, one step = precalc()+step()[2 sub-steps]+density()"
                            % (spec["nr"], spec["nz"], n_local),
                "particles_per_gpu": n_local, "grid": [spec["nr"], spec["nz"]],
                "parallelism": "particle shards x%d (%s), replicated grid, one all-reduce of the per-cell sums per frame (%s)"
                               % (world, "fixed total of %d" % n_total if strong else "fixed per GPU",
                                  "library RCCL communicator" if args.comm == "lib" else "torch.distributed: " + args.comm),
            },
            "comm": comm,
            "roofline": {
                "bound": "hbm",
                "kernel": "push_tiles_kernel<float, fuse, rebin> (step(): K3+K1+K2 x2 sub-steps, with the scatter's "
                          "per-cell sums, the tile census and every ~4th launch the re-binning fused in)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_UPDATE * 2.0 * n_local,
                "avg_launch_ms": push_ms,
                "launches_timed": st["step_launches"],
                "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS,
                # north_star's wording: "% of HBM-read roofline for the push" — only the 24 B an update must READ (6 scalars,
                # SURVEY 8(d)) over the peak: the same launch time priced at half the bytes
                "frac_of_hbm_read_roofline": 0.5 * achieved / HBM_PEAK_GBS,
                "hbm_read_roofline_updates_per_s": HBM_PEAK_GBS * 1e9 / 24.0,
            },
            "kernel_ms_per_step": {
                "push_incl_fused_scatter_and_rebinning": st["ms_push"] / args.steps,
                "separate_cell_sums": st["ms_deposit"] / args.steps,
                "stamp_normalise_ema": st["ms_stamp"] / args.steps, "precalc": st["ms_precalc"] / args.steps,
                "bin_table_scan": st["ms_sort"] / args.steps, "rebinning_launches": st["sort_passes"],
                "last_spilled": st["deposit_spilled"],
            },
            "setup": {"first_binning_ms": first_binning_ms,
                      "what": "count + scan + two-level scatter of the randomly ordered upload (host clock, synchronised), outside the timed region"},
        }
        out["roofline"]["cycle_frac_of_hbm_read_roofline"] = out["value"] / world * 24.0 / (HBM_PEAK_GBS * 1e9)
        if comm_fallback:
            out["comm_fallback"] = comm_fallback
        if rehearsal_note:
            out["rehearsal"] = rehearsal_note
        tr = measured_traffic({"particles_per_gpu": n_local, "grid": [spec["nr"], spec["nz"]], "rng": args.rng, "dtype": "f32"})
        if tr is None:
            out["roofline"]["traffic_source"] = "no committed PMC pass for this configuration (profiles/r*_traffic.json)"
        if tr and world == 1:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_source"] = "%s: %s" % (tr.get("file"), tr.get("source"))
            if push_ms > 0 and tr.get("bytes_per_launch"):
                # what the kernel really moves per second (PMC bytes / measured launch time): the excess over
                # `achieved` is the reference RNG's entropy-table gather (64-B lines for 16-B texels), DESIGN.md 4.2
                rate = tr["bytes_per_launch"] / (push_ms * 1e-3) / 1e9
                out["roofline"]["traffic_rate_GBs"] = rate
                out["roofline"]["traffic_rate_frac_of_peak"] = rate / HBM_PEAK_GBS
        if args.rng == "counter":
            out["config"]["workload"] += "; EXTENSION: counter-based RNG (Philox4x32-10) instead of the reference's entropy-table generator"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]

    sim.destroy()
    if rank == 0 and world == 1 and not distributed and args.rng == "reference" and not args.no_extensions:
        # Extension, reported beside the headline and never as it: same scene, same frame, with
        # the counter-based generator (no entropy table, no per-particle random state).
        ext = build("counter")
        ext.precalc(); ext.sort()
        for _ in range(args.warmup):
            ext.precalc(); ext.step(); ext.density()
        ext.sync(); torch.cuda.synchronize()
        ext.resetStats(); ext.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ext.precalc(); ext.step(); ext.density()
        ext.sync(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        es = ext.stats()
        pm = es["ms_push"] / max(1, es["step_launches"])
        out["extensions"] = {"counter_rng": {
            "what": "spec.rng_mode = 1: Philox4x32-10(particle id, sub-step) replaces the reference's entropy-table walk; "
                    "parity is bit-exact against the oracle's counter mode, not against the reference's random stream",
            "value": 2.0 * n_local * args.steps / el, "unit": "particle-updates/s", "ms_per_step": 1e3 * el / args.steps,
            "push_avg_launch_ms": pm, "cell_sums_avg_launch_ms": es["ms_deposit"] / max(1, es["deposit_launches"]),
            "rebinning_launches": es["sort_passes"],
            # Two accountings, named for what they are.  (1) SURVEY 8(d)'s 48 B per update x the two updates of a
            # launch: a THROUGHPUT figure comparable with the headline's roofline.frac, not bus utilisation — the launch fuses two
            # sub-steps and moves each particle once.  (2) What the launch really streams: six coordinates read + written and
            # the alive flag, 50 B per particle and launch (no per-particle random state in this mode), over its time and the peak.
            "accounting_48B_per_update_GBs": ALGO_BYTES_PER_UPDATE * 2.0 * n_local / (pm * 1e-3) / 1e9,
            "accounting_48B_per_update_frac": ALGO_BYTES_PER_UPDATE * 2.0 * n_local / (pm * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "streamed_bytes_per_launch": 50.0 * n_local,
            "streamed_GBs": 50.0 * n_local / (pm * 1e-3) / 1e9,
            "streamed_frac": 50.0 * n_local / (pm * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }}
        ext.destroy()
        # "C2 with both cic and ref11 reported": the same frame with the bilinear deposit (extension key shape:'cic')
        cic = build("reference", shape="cic")
        cic.precalc(); cic.sort()
        for _ in range(args.warmup):
            cic.precalc(); cic.step(); cic.density()
        cic.sync(); torch.cuda.synchronize()
        cic.resetStats(); cic.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cic.precalc(); cic.step(); cic.density()
        cic.sync(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        cs = cic.stats()
        out["extensions"]["c2_cic"] = {
            "what": "spec.shape = 1 (extension, no reference counterpart): density() deposits bilinearly on the four nearest cell centres "
                    "(16 LDS atomics per particle in a separate pass) instead of the reference's 11x11 sprite (4, fused into the push)",
            "value": 2.0 * n_local * args.steps / el, "unit": "particle-updates/s", "ms_per_step": 1e3 * el / args.steps,
            "push_avg_launch_ms": cs["ms_push"] / max(1, cs["step_launches"]),
            "cic_sums_avg_launch_ms": cs["ms_deposit"] / max(1, cs["deposit_launches"]), "rebinning_launches": cs["sort_passes"]}
        cic.destroy()
        torch.cuda.empty_cache()
        out["extensions"]["node_host"] = node_host_line(args.side, args.grid, args.steps, args.warmup, out["ms_per_step"])
        out["extensions"]["dense_sor"] = dense_sor_line(local_rank)
        out["extensions"]["em"] = em_line(local_rank, args.c3_particles, args.c3_grid, max(2, args.steps // 4), 1, "fp64")
        out["extensions"]["c3"] = es3d_line(local_rank, args.c3_particles, args.c3_grid, max(8, args.steps // 4), 1, stream=stream,
                                            cpu=not args.no_cpu_baseline)
        if not args.no_c5:
            # BASELINE configs[4] at its stated size — 512^3 Yee lattice, 2e9 electrons, fp64 — on this ONE GPU (233 of its 288 GB):
            # the 8-GPU form of that configuration is a decomposition of exactly this.  A report, never a reason to lose
            # the lines above: whatever goes wrong is recorded in its place.
            try:
                torch.cuda.empty_cache()
                free = torch.cuda.mem_get_info(local_rank)[0]
                if free < 250e9:
                    raise RuntimeError("%.0f GB free on the device, 250 asked for" % (free / 1e9))
                out["extensions"]["c5_one_gpu"] = em_line(local_rank, 2_000_000_000, 512, 2, 1, "fp64")
            except Exception as e:
                out["extensions"]["c5_one_gpu"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
    if not args.no_strong_c4:
        # north_star's scaling target lives on another workload than the parity-pinned headline: BASELINE configs[3]
        # (512^3 nodes, 2e9 particles, two species) as ONE decomposed run over the launcher's ranks — a fixed total, so
        # N = 1, 2, 4, 8 of the driver's plain `bench.py --gpus N` trace the strong-scaling curve in this block, over the
        # library's own RCCL communicator (there is no other transport here: a failure to set it up ends the run).
        torch.cuda.empty_cache()
        # A rank that fails inside this block leaves the others waiting in a collective of the library, rank 0 possibly
        # among them: the headline measured above must not be lost with it.  A watchdog on every rank ends the wait: rank 0
        # prints the line with the block marked as failed, every rank leaves.  
        import threading
        limit = float(os.environ.get("FPIC_BENCH_C4_LIMIT_S", "420"))
        headline = json.dumps(out) if rank == 0 else None   # (the timer's thread prints a snapshot, not the dictionary the main thread holds)

        def give_up():
            if rank == 0:
                line = json.loads(headline)
                line["strong_c4"] = {"value": None, "error": "the strong_c4 block did not finish within %.0f s on rank 0 (a rank failed or an exchange "
                                                             "never completed); the headline above was measured before it" % limit}
                print(json.dumps(line), flush=True)
            sys.stderr.write("bench.py: rank %d gives up on the strong_c4 block after %.0f s\n" % (rank, limit))
            sys.stderr.flush()
            os._exit(5)  # every rank, rank 0 included: a block that hung is a failed run, as on the exception path below (the line is on stdout)

        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            block = box_workload(args, rank, world, local_rank, dist if distributed else None, steps=max(2, args.steps // 5), warmup=1,
                                 cpu=not args.no_cpu_baseline)
            watchdog.cancel()
        except Exception as e:
            watchdog.cancel()  # loud, but the headline measured above is not lost with it: the line, then a non-zero exit
            if rank == 0:
                out["strong_c4"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
                print(json.dumps(out), flush=True)
            sys.stderr.write("bench.py: the strong_c4 block failed on rank %d: %r\n" % (rank, e))
            sys.stderr.flush()
            os._exit(4)  # (the other ranks may sit in a collective of the library: the launcher ends them)
        if rank == 0:
            block["what"] = ("north_star's strong-scaling workload (BASELINE configs[3]) measured in the same run at this N: value = total "
                             "particle-updates/s of the whole decomposed job; the driver's ratio value(N)/value(1) of THIS block is the >= 6x curve")
            out["strong_c4"] = block
    if rank == 0:
        print(json.dumps(out), flush=True)

    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when a communicator
    # is created): file descriptor 1 is pointed at stderr for the whole run, and Python's own stdout — the JSON line —
    # keeps the real one.
    sys.stdout.flush()
    _real = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(_real, "w", buffering=1)
    main()
