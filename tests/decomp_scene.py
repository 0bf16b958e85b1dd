"""Shared by the drivers of tests/test_gpu_fake_rccl.py (ranks = threads of one process) and
tests/test_gpu_rccl_processes.py (ranks = processes): one seeded scene of the CART3D box, the one-handle run it is
compared with, one rank's run through the library's communicator + decomposition, and the comparison.  Every process
rebuilds the scene from the same seed, so nothing but the unique id and the case travels between them."""
import numpy as np

C = 2.998e8


def build(fp, case):
    world, shape, G, every = case["world"], tuple(case["shape"]), case["ghost"], case["every"]
    em = case["em"]
    rng = np.random.default_rng(case["seed"])
    n = case["n"]
    L = tuple(1e-3 * s for s in shape)
    if em:
        d = [L[a] / shape[a] for a in range(3)]
        dt = 0.5 / (C * np.sqrt(sum(1 / x ** 2 for x in d)))
    else:
        dt = 5e-12
    spec = dict(radius=L[0], length_y=L[1], height=L[2], nr=shape[0], ny=shape[1], nz=shape[2], dt=dt, nparticles=0, count=n,
                particle_mass=9.109e-31, particle_charge=-1.602e-19, geometry="cart3d", solver="yee" if em else "poisson_fft",
                macro_weight=(1e9 if case.get("emptying") else 1e15) * np.prod(L) / n)   # (a beam that must not blow itself beyond the ghost planes)
    nzl = shape[2] // world
    pos = rng.random((n, 3)) * L
    vz = 0.7 * G * 1e-3 / (every * dt * C)
    vel = np.stack([rng.normal(0, 0.05, n), rng.normal(0, 0.05, n), rng.uniform(-min(vz, 0.9), min(vz, 0.9), n)], axis=1)
    if case.get("emptying"):  # everything starts in slab 0 and streams upwards: rank 0 empties, the others fill
        pos[:, 2] = (0.1 + 0.8 * rng.random(n)) * L[2] / world
        vel[:, 2] = min(vz, 0.9)
    owner = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(owner, kind="stable")
    pos, vel, counts = pos[order], vel[order], np.bincount(owner, minlength=world)
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))
    dist_solve = case["distributed_solve"]
    # (the library's own transforms on power-of-two grids: the decomposed solve is the one handle's, bit for bit)
    own_fft = all(s_ & (s_ - 1) == 0 and s_ >= 8 for s_ in shape)
    fields = [fp.F3_J_FIXED, fp.F3_EDGE_E, fp.F3_FACE_B] if em else [fp.F3_RHO_FIXED] + ([] if (dist_solve and not own_fft) or dist_solve == 2 else [fp.F3_E])
    return dict(case=case, world=world, shape=shape, G=G, every=every, em=em, dist_solve=dist_solve, precision=case["precision"], n=n, spec=spec, nzl=nzl,
                pos=pos, vel=vel, counts=counts, E=E, B=B, fields=fields, frames=case.get("frames", 4), em_from_precalc=bool(em and case.get("precalc")))


def _start(sc, s):
    if sc["em"] and not sc["em_from_precalc"]:
        s.set(edge_E=sc["E"], face_B=sc["B"])
    else:
        if sc["em"]:
            s.addB(0.0, 0.0, 0.01)
        s.precalc()


def run_one(fp, sc):
    one = fp.makeCylindricalParticlePusher(sc["spec"], precision=sc["precision"])
    one.set(position=sc["pos"], velocity=sc["vel"])
    _start(sc, one)
    for _ in range(sc["frames"]):
        one.step()
    ref = one.getParticles()
    ref_f = {w: one.readField(w).reshape(sc["shape"][2], -1) for w in sc["fields"]}
    one.destroy()
    return ref, ref_f


def run_rank(fp, sc, r, uid):
    """rank r of the world, through fpic_comm_init + fpic_domain_init + fpic_precalc / fpic_step: the code a multi-GPU job runs"""
    world, nzl, counts = sc["world"], sc["nzl"], sc["counts"]
    s = fp.makeCylindricalParticlePusher(dict(sc["spec"], count=3 * sc["n"]), precision=sc["precision"])
    s.commInit(uid, r, world)
    s.domainInit(r, world, ghost_planes=sc["G"], migrate_every=sc["every"], distributed_solve=sc["dist_solve"])
    first = int(counts[:r].sum())
    s.domainSet(sc["pos"][first:first + counts[r]], sc["vel"][first:first + counts[r]], first_id=first)
    _start(sc, s)
    for _ in range(sc["frames"]):
        s.step()
    got = s.domainGet()
    out = (got, {w: s.readField(w).reshape(sc["shape"][2], -1)[r * nzl:(r + 1) * nzl].copy() for w in sc["fields"]}, s.domainStats(), len(got["ids"]))
    s.destroy()
    return out


def same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def compare(fp, sc, ref, ref_f, out):
    world, nzl, n, fields = sc["world"], sc["nzl"], sc["n"], sc["fields"]
    ids = np.concatenate([o[0]["ids"] for o in out])
    idx = np.argsort(ids, kind="stable")
    pos = np.concatenate([o[0]["position"] for o in out])[idx]
    vel = np.concatenate([o[0]["velocity"] for o in out])[idx]
    res = {"ids_ok": bool(np.array_equal(ids[idx], np.arange(n))), "pos_same": bool(same(pos, ref["position"])), "vel_same": bool(same(vel, ref["velocity"])),
           "migrated": int(sum(o[2]["migrated"] for o in out)), "lost": int(sum(o[2]["lost"] for o in out)), "fields": {}, "held": [int(o[3]) for o in out]}
    if pos.shape == ref["position"].shape:
        d = np.abs(pos.astype(np.float64) - ref["position"].astype(np.float64))
        d = np.minimum(d, 1 - d)
        res["pos_err"] = float(d.max())
    else:
        res["pos_err"] = float("inf")
    for w in fields:
        res["fields"][str(w)] = all(bool(same(out[r][1][w], ref_f[w][r * nzl:(r + 1) * nzl])) for r in range(world))
    # the integer charge grid of all ranks' own planes: its total is exact whatever the solve's summation order did to the particles
    own = np.concatenate([out[r][1][fields[0]] for r in range(world)]).astype(np.int64)
    if fields[0] == fp.F3_RHO_FIXED:
        want = ref_f[fields[0]].astype(np.int64)
        res["charge_total_same"] = bool(int(own.sum()) == int(want.sum()))
        res["charge_max_rel_diff"] = float(np.abs(own - want).max() / max(1, np.abs(want).max()))
    else:
        res["charge_total_same"], res["charge_max_rel_diff"] = True, 0.0
    return res
