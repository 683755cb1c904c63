"""Spatial domain decomposition of the SPH hot path over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).  The
reference has no distributed code at all (only SLURM replicas), so this is new
design, following SURVEY.md 8(e):

  * recursive coordinate bisection of the particle set into `world` axis-aligned regions with
    equal counts; a particle is OWNED by the rank whose region contains it;
  * per step, each rank imports GHOST copies of the foreign particles it can need: every rank
    publishes, on a coarse global grid, how far the search radii of its owned particles reach
    (a dilated "need map"), and a particle is sent to every rank whose map covers its cell - so
    the halo is as thick as the LOCAL kNN radius, thin in dense regions; the rank searches / sums
    over owned+ghost candidates for its owned particles only, and refreshes the ghosts' h_j,
    rho_j and m*Pi_j from their owners between the passes (the kernel uses the NEIGHBOUR's h
    (nsc:587-588), rho (nsc:646) and Pi (nsc:651)) - four point-to-point halo phases;
  * exchange = grouped isend/irecv between the (at most world-1) peers that share a halo, no
    collective on the data path; the collectives are the need maps and counts of a replan and
    ONE scalar reduction per step (all_reduce(max) of [halo verdict, -crossing time]: dt is the
    global minimum crossing time, nsc:786);
  * the send lists (the "plan") are reused from step to step with a skin, molecular-dynamics
    style, while displacements since the last plan stay below half the skin; under the
    reference's large dt they rarely do, so replanning is kept cheap (fused need-map kernel,
    migration of strays only on every 4th replan, staleness checks skipped while every step
    replans anyway).

Halo coverage is verified every step (h_i + 2 x displacement <= claimed reach for every owned
particle, otherwise all ranks replan with a wider halo and redo the step's search and sums), so
the decomposed result equals the single-GPU result.

Compute is injected through a small backend interface; the product backend `LibBackend` calls
libsphx.so's sphx_dev_* entry points on the rank's GPU (no CPU fallback).
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HUGE_CT = float("inf")      # "no gas particle voted a crossing time" (SPHX_CT_NONE: votes are at most DBL_MAX)


# ==============================================================================================
# decomposition
# ==============================================================================================
def rcb_regions(points, world):
    """Recursive coordinate bisection -> (owner (N,) int64, lo (world,3), hi (world,3)).
    Regions tile all of space (outer faces at -inf/+inf); counts differ by at most 1 per cut."""
    pts = np.asarray(points, dtype=np.float64)
    n = len(pts)
    owner = np.zeros(n, np.int64)
    lo = np.full((world, 3), -np.inf)
    hi = np.full((world, 3), np.inf)

    def split(idx, r0, r1, blo, bhi):
        nr = r1 - r0
        if nr == 1:
            owner[idx] = r0
            lo[r0], hi[r0] = blo, bhi
            return
        nl = nr // 2
        sub = pts[idx]
        ext = sub.max(axis=0) - sub.min(axis=0) if len(idx) else np.zeros(3)
        ax = int(np.argmax(ext))
        kcut = int(round(len(idx) * nl / nr))
        kcut = min(max(kcut, 1), max(len(idx) - 1, 1))
        order = np.argpartition(sub[:, ax], kcut - 1) if len(idx) > 1 else np.arange(len(idx))
        left_vals = sub[order[:kcut], ax]
        right_vals = sub[order[kcut:], ax]
        cut = 0.5 * (left_vals.max() + right_vals.min()) if len(right_vals) and len(left_vals) else 0.0
        lhi = np.array(bhi); lhi[ax] = cut
        rlo = np.array(blo); rlo[ax] = cut
        split(idx[order[:kcut]], r0, r0 + nl, np.array(blo), lhi)
        split(idx[order[kcut:]], r0 + nl, r1, rlo, np.array(bhi))

    split(np.arange(n), 0, world, np.full(3, -np.inf), np.full(3, np.inf))
    return owner, lo, hi


def brick_regions(world, box_lo, box_hi):
    """The regions rcb_regions would cut a UNIFORM fill of the box [box_lo, box_hi] into, without the particles: recursive
    bisection of the box along its longest edge, volumes in proportion to the ranks on each side.  -> (lo, hi) (world,3)
    with the outer faces at -inf / +inf like rcb_regions', and the finite boxes (blo, bhi) to fill."""
    lo = np.full((world, 3), -np.inf); hi = np.full((world, 3), np.inf)
    flo = np.zeros((world, 3)); fhi = np.zeros((world, 3))

    def split(r0, r1, blo, bhi, glo, ghi):
        nr = r1 - r0
        if nr == 1:
            lo[r0], hi[r0], flo[r0], fhi[r0] = glo, ghi, blo, bhi
            return
        nl = nr // 2
        ax = int(np.argmax(bhi - blo))
        cut = blo[ax] + (bhi[ax] - blo[ax]) * nl / nr
        lhi, rlo = np.array(bhi), np.array(blo)
        lhi[ax] = cut; rlo[ax] = cut
        glhi, grlo = np.array(ghi), np.array(glo)
        glhi[ax] = cut; grlo[ax] = cut
        split(r0, r0 + nl, np.array(blo), lhi, np.array(glo), glhi)
        split(r0 + nl, r1, rlo, np.array(bhi), grlo, np.array(ghi))

    split(0, world, np.asarray(box_lo, dtype=np.float64), np.asarray(box_hi, dtype=np.float64),
          np.full(3, -np.inf), np.full(3, np.inf))
    return lo, hi, flo, fhi


def region_of(pos, lo, hi):
    """Owner rank of each position (torch (n,3)); regions tile space, ties go to the lower rank."""
    inside = ((pos[:, None, :] >= lo[None]) & (pos[:, None, :] <= hi[None])).all(dim=2)   # (n, world)
    return torch.argmax(inside.to(torch.int8), dim=1)


def dist_to_region(pos, lo, hi):
    """Euclidean distance from each position (n,3) to the box [lo,hi] (infinite faces count 0)."""
    d = torch.clamp(lo[None] - pos, min=0.) + torch.clamp(pos - hi[None], min=0.)
    d = torch.nan_to_num(d, nan=0.0, posinf=0.0, neginf=0.0)
    return torch.sqrt((d * d).sum(dim=1))


# ==============================================================================================
# point-to-point halo exchange
# ==============================================================================================
class Exchanger:
    """Variable-size row exchange between all ranks with grouped isend/irecv."""

    def __init__(self, rank, world, comm_device):
        self.rank, self.world, self.comm_device = rank, world, comm_device
        self.bytes_sent = 0

    def warm_up(self):
        """One element to and from every peer: whatever a backend sets up lazily per pair of ranks (RCCL:
        a communicator per pair, tens to hundreds of ms) happens here and not when a particle first
        migrates to a rank this one had no halo with."""
        if self.world == 1:
            return
        ops, keep = [], []
        for p in range(self.world):
            if p == self.rank:
                continue
            t = torch.full((1,), float(self.rank), dtype=torch.float64, device=self.comm_device)
            r = torch.empty(1, dtype=torch.float64, device=self.comm_device)
            ops += [dist.P2POp(dist.isend, t, p), dist.P2POp(dist.irecv, r, p)]
            keep += [(p, r), (None, t)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for p, r in keep:
            if p is not None and float(r[0]) != float(p):
                raise RuntimeError("halo warm-up: rank %d got %r from rank %d" % (self.rank, float(r[0]), p))

    def rows(self, send_bufs, recv_counts, width, dtype=torch.float64, into=None, defer=False):
        """send_bufs[p]: (count_p, width) tensor for peer p (any device) -> list of received
        (recv_counts[p], width) tensors on the compute device of send_bufs.  `into`: a contiguous
        (sum(recv_counts), width) tensor on the communication device; peers' rows are then received
        straight into its consecutive slices (which the returned list aliases).  defer=True: the transfers are
        started and a function is returned that completes them (and gives the list): what the caller launches in
        between runs beside the exchange (RCCL moves the rows on its own stream; wait() orders the streams)."""
        out = [None] * self.world
        if self.world == 1:
            return (lambda: out) if defer else out
        ops, stage = [], []
        dev = None
        o = 0
        for p in range(self.world):
            if p == self.rank:
                continue
            sb = send_bufs[p]
            dev = sb.device if dev is None else dev
            if sb.shape[0] > 0:
                t = sb.contiguous().to(self.comm_device)
                self.bytes_sent += t.numel() * t.element_size()
                ops.append(dist.P2POp(dist.isend, t, p))
                stage.append(t)
            if recv_counts[p] > 0:
                if into is not None:
                    r = into[o:o + recv_counts[p]]
                    o += recv_counts[p]
                else:
                    r = torch.empty((recv_counts[p], width), dtype=dtype, device=self.comm_device)
                ops.append(dist.P2POp(dist.irecv, r, p))
                out[p] = r
        works = dist.batch_isend_irecv(ops) if ops else []

        def finish():
            for w in works:
                w.wait()
            stage.clear()
            for p in range(self.world):
                if out[p] is not None and dev is not None and out[p].device != dev:
                    out[p] = out[p].to(dev)
            return out
        return finish if defer else finish()


# ==============================================================================================
# compute backend on the GPU: libsphx sphx_dev_* (include/sphx.h)
# ==============================================================================================
class LibBackend:
    def __init__(self, device_index, k=40, dist_bound=0.0, clip_grad=False):
        from . import _lib
        self.ctx = _lib.Context(device_index)
        self.ctx.check(self.ctx.lib.sphx_set_clip_grad(self.ctx.h, 1 if clip_grad else 0))
        self.lib = self.ctx.lib
        self.k = k
        self.dist_bound = dist_bound
        self.device = torch.device("cuda", device_index)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.check(self.lib.sphx_set_stream(self.ctx.h, C.c_void_p(stream)))
        self.n_total = 0

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _chk(self, rc):
        self.ctx.check(rc)

    def clamp(self, pos, vel):
        self._chk(self.lib.sphx_dev_clamp(self.ctx.h, pos.shape[0], self._p(pos), self._p(vel)))

    @staticmethod
    def _table(tensors):
        arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        wid = (C.c_int32 * len(tensors))(*[1 if t.dim() == 1 else int(t.shape[1]) for t in tensors])
        return arr, wid

    def pack_rows(self, idx, fields):
        """rows[t] = concatenated fields of particle idx[t] (one launch; include/sphx.h sphx_dev_pack_rows)."""
        fields = [f.contiguous() for f in fields]
        n = int(fields[0].shape[0] if idx is None else idx.numel())
        W = sum(1 if f.dim() == 1 else int(f.shape[1]) for f in fields)
        rows = torch.empty((n, W), dtype=torch.float64, device=self.device)
        if n:
            arr, wid = self._table(fields)
            self._chk(self.lib.sphx_dev_pack_rows(self.ctx.h, n, self._p(idx), len(fields), arr, wid, self._p(rows)))
        return rows

    SLACK = 0.3      # room behind a rank's owned arrays for its ghosts (append_rows), as a fraction of the owned count

    def regroup(self, sel, n_sel, rows, fields):
        """New arrays: fields[sel] followed by the particles in `rows` (one launch; sphx_dev_regroup).  Each comes with
        room behind it (SLACK), so that append_rows can put the step's ghosts there without copying the owned part."""
        fields = [f.contiguous() for f in fields]
        nr = 0 if rows is None else int(rows.shape[0])
        n = n_sel + nr
        cap = n + int(self.SLACK * n) + 1024
        outs = [torch.empty((cap,) + tuple(f.shape[1:]), dtype=f.dtype, device=self.device)[:n] for f in fields]
        if n:
            arr, wid = self._table(fields)
            oarr, _ = self._table(outs)
            self._chk(self.lib.sphx_dev_regroup(self.ctx.h, n_sel, self._p(sel), nr,
                                                self._p(rows.contiguous()) if nr else None, len(fields), arr, wid, oarr))
        return outs

    def append_rows(self, n_owned, rows, fields):
        """The owned arrays `fields` ((n_owned,) or (n_owned, w), float64) with the particles in `rows` appended IN PLACE:
        -> (complete, owned), complete[q] = the (n_owned + len(rows), ...) array whose first n_owned entries are the
        storage of owned[q].  Where a field's storage has room behind the owned part (arrays made by regroup, or by an
        earlier call) nothing but the new rows is written; otherwise the field moves once to a roomier buffer and
        owned[q] is the new owned view, which the caller keeps in place of the old one."""
        nr = 0 if rows is None else int(rows.shape[0])
        need = n_owned + nr
        complete, owned = [], []
        for f in fields:
            w = 1 if f.dim() == 1 else int(f.shape[1])
            st = f.untyped_storage()
            room = (st.nbytes() // f.element_size() - f.storage_offset()) // w
            if f.is_contiguous() and int(f.shape[0]) == n_owned and room >= need:
                full = torch.empty(0, dtype=f.dtype, device=f.device).set_(st, f.storage_offset(), (need,) + tuple(f.shape[1:]))
            else:
                buf = torch.empty((need + int(self.SLACK * need) + 1024,) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
                buf[:n_owned].copy_(f)
                f, full = buf[:n_owned], buf[:need]
            complete.append(full)
            owned.append(f)
        if nr:
            tails = [c[n_owned:] for c in complete]
            oarr, wid = self._table(tails)
            self._chk(self.lib.sphx_dev_regroup(self.ctx.h, 0, None, nr, self._p(rows.contiguous()), len(fields), None, wid, oarr))
        return complete, owned

    def need_map(self, pos, w, g_lo, g_cs, G):
        """uint8 (G^3,) map of the coarse cells a neighbour of an owned particle can lie in (one launch;
        include/sphx.h sphx_dev_need_map).  g_lo: 3 host floats."""
        out = torch.empty(G * G * G, dtype=torch.uint8, device=self.device)
        lo = (C.c_double * 3)(*g_lo)
        self._chk(self.lib.sphx_dev_need_map(self.ctx.h, int(pos.shape[0]), self._p(pos.contiguous()),
                                             self._p(w.contiguous()), lo, float(g_cs), int(G), self._p(out)))
        return out

    def set_reach_cap(self, cap_abs):
        self._chk(self.lib.sphx_dev_set_reach_cap(self.ctx.h, float(cap_abs)))

    def reach(self, h, vel, halo_scale, skin_frac, dt_last):
        """w_i = max((halo + skin) h_i, halo h_i + |v_i| dt) in one launch (sphx_dev_reach)."""
        w = torch.empty_like(h)
        self._chk(self.lib.sphx_dev_reach(self.ctx.h, int(h.shape[0]), self._p(h.contiguous()), self._p(vel.contiguous()),
                                          float(halo_scale), float(skin_frac), float(dt_last), self._p(w)))
        return w

    def reach_dt(self, h, vel, halo_scale, skin_frac, dt_dev):
        """sphx_dev_reach with dt taken from device memory (the (1,) tensor sphx_dev_integrate_auto / _loop filled)."""
        w = torch.empty_like(h)
        self._chk(self.lib.sphx_dev_reach_dt(self.ctx.h, int(h.shape[0]), self._p(h.contiguous()), self._p(vel.contiguous()),
                                             float(halo_scale), float(skin_frac), self._p(dt_dev), self._p(w)))
        return w

    def plan_mask(self, pos, g_lo, g_cs, G, maps, rank):
        """(mask (W, n) uint8, counts (W,) int64): which owned particle goes to which peer as a ghost, and how many, from
        the ranks' need maps (W, G^3) - one launch (sphx_dev_plan_mask)."""
        W, n = int(maps.shape[0]), int(pos.shape[0])
        mask = torch.empty((W, n), dtype=torch.uint8, device=self.device)
        counts = torch.empty(W, dtype=torch.int64, device=self.device)
        lo = (C.c_double * 3)(*g_lo)
        self._chk(self.lib.sphx_dev_plan_mask(self.ctx.h, n, self._p(pos.contiguous()), lo, float(g_cs), int(G), W, int(rank),
                                              self._p(maps.contiguous()), self._p(mask), self._p(counts)))
        return mask, counts

    def step_scalars(self, n_owned, h, w_plan, D, hclip, ct):
        """(4,) device tensor {halo too thin?, -crossing time, max h, clipped mean h} in one launch
        (sphx_dev_step_scalars)."""
        out = torch.empty(4, dtype=torch.float64, device=self.device)
        self._chk(self.lib.sphx_dev_step_scalars(self.ctx.h, int(n_owned), self._p(h), self._p(w_plan), float(D),
                                                 float(hclip), self._p(ct), self._p(out)))
        return out

    def search(self, pos, n_owned, hint, mean_h):
        n = pos.shape[0]
        self.n_total = n
        h = torch.zeros(n, dtype=torch.float64, device=pos.device)      # ghosts' entries stay 0 until their halo
        self._chk(self.lib.sphx_dev_set_mean_h(self.ctx.h, float(mean_h)))
        self._chk(self.lib.sphx_dev_search(self.ctx.h, n, n_owned, self.k, self._p(pos), self._p(hint), 0.0,
                                           float(self.dist_bound), self._p(h)))
        return h

    def order(self, n_total):
        """Cell order of the last search (sorted position -> caller index), int64 tensor."""
        out = torch.empty(n_total, dtype=torch.int32, device=self.device)
        self._chk(self.lib.sphx_dev_get_order(self.ctx.h, n_total, self._p(out)))
        return out.to(torch.int64)

    def prep(self, pos, vel, m, h, T, mu, gam, ptype):
        self._keep = (pos, vel, m, h, T, mu, gam, ptype)
        self._chk(self.lib.sphx_dev_prep(self.ctx.h, *[self._p(t) for t in self._keep]))

    def select_blobs(self, part):
        """The passes that follow take: 1 the interior blobs (no ghost among their particles' neighbours; prep: the owned
        particles' records), 2 the boundary blobs (prep: the ghosts'), 0 everything.  True if in force (sphx_dev_select_blobs);
        a pass is then called once per part with the same output arrays (out=)."""
        rc = self.lib.sphx_dev_select_blobs(self.ctx.h, int(part))
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def density(self, want_dust=False, out=None):
        n, dev = self.n_total, self.device
        # the passes write every owned particle's entry straight into these (no copy); ghosts' entries are
        # either filled by the halo exchange (rho, m Pi) or never read
        if out is None:
            rho = torch.empty(n, dtype=torch.float64, device=dev)
            nden = torch.empty(n, dtype=torch.float64, device=dev)
            ha = torch.empty((n, 3), dtype=torch.float64, device=dev)
            self.rhod = torch.empty(n, dtype=torch.float64, device=dev) if want_dust else None
        else:
            rho, nden, ha = out
        self._chk(self.lib.sphx_dev_density(self.ctx.h, self._p(rho), self._p(self.rhod), self._p(nden), self._p(ha)))
        return rho, nden, ha

    def pi(self, rho_complete, out=None):
        n, dev = self.n_total, self.device
        if out is None:
            bw = torch.empty(n, dtype=torch.float64, device=dev)
            ct = torch.zeros(1, dtype=torch.float64, device=dev)
        else:
            bw, ct = out
        self._chk(self.lib.sphx_dev_pi(self.ctx.h, self._p(rho_complete), None, self._p(bw), self._p(ct)))
        return bw, ct

    def visc(self, bw_complete, m, out=None):
        n, dev = self.n_total, self.device
        if out is None:
            va = torch.empty((n, 3), dtype=torch.float64, device=dev)
            vh = torch.empty(n, dtype=torch.float64, device=dev)
        else:
            va, vh = out
        self._chk(self.lib.sphx_dev_visc(self.ctx.h, self._p(bw_complete), self._p(m), self._p(va), self._p(vh)))
        return va, vh

    def integrate_auto(self, n_owned, pos, vel, acc, E, T, m, mu, gam, ptype, ha, va, vh, red2, first, fixed_dt):
        """Leapfrog with verdict and dt taken on the device (sphx_dev_integrate_auto) -> dt as a (1,) device tensor
        (0 when the verdict says the step must be redone: nothing was changed then)."""
        dt = torch.empty(1, dtype=torch.float64, device=self.device)
        self._chk(self.lib.sphx_dev_integrate_auto(self.ctx.h, n_owned, self._p(pos), self._p(vel), self._p(acc),
                                                   self._p(E), self._p(T), self._p(m), self._p(mu), self._p(gam),
                                                   self._p(ptype), self._p(ha), self._p(va), self._p(vh),
                                                   self._p(red2), 1 if first else 0, float(fixed_dt), self._p(dt)))
        return dt

    # ---- gas-dust drag (nsc:719-742) and the species pass (nsc:624-627, + metallicity and AGB yields) ----
    def drag(self, m, ptype, mgm, mcs):
        """-> (drag_on (n_total,3): owned rows; drag_reaction (n_total,3): every row, ghosts' rows go back to their owners)."""
        n, dev = self.n_total, self.device
        onto = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        react = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        self._keep_drag = (m, ptype, mgm, mcs)
        self._chk(self.lib.sphx_dev_drag(self.ctx.h, self._p(m), self._p(ptype), self._p(mgm), self._p(mcs), self._p(onto),
                                         self._p(react)))
        return onto, react

    def set_drag_terms(self, onto, react, rho, rhod):
        self._keep_terms = (onto.contiguous(), react.contiguous(), rho.contiguous(), rhod.contiguous())
        self._chk(self.lib.sphx_dev_set_drag_terms(self.ctx.h, *[self._p(t) for t in self._keep_terms]))

    def set_agb(self, nspecies, table, mu_specie, solar_mass):
        splines, mapto, divisor = table
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        ntx = i32([sp.get_knots()[0].size for sp in splines]); nty = i32([sp.get_knots()[1].size for sp in splines])
        cat = lambda parts: f64(np.concatenate([np.asarray(q, dtype=np.float64).ravel() for q in parts]))
        tx = cat([sp.get_knots()[0] for sp in splines]); ty = cat([sp.get_knots()[1] for sp in splines])
        cf = cat([sp.get_coeffs() for sp in splines]); mp = i32(mapto); mu = f64(mu_specie[:nspecies])
        p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        self._chk(self.lib.sphx_dev_set_agb(self.ctx.h, int(nspecies), len(splines), p32(ntx), p32(nty), dp(tx), dp(ty), dp(cf),
                                            p32(mp), float(divisor), dp(mu), float(solar_mass)))
        self.has_agb = True

    def species(self, fun, m):
        """-> (F (S,n_total), Z (n_total,) | None, agb_dust (n_total,S) | None); owned entries are meaningful."""
        n, dev = self.n_total, self.device
        S = int(fun.shape[1])
        F = torch.zeros((S, n), dtype=torch.float64, device=dev)
        Z = A = None
        if getattr(self, "has_agb", False):
            Z = torch.zeros(n, dtype=torch.float64, device=dev)
            A = torch.zeros((n, S), dtype=torch.float64, device=dev)
        fun = fun.contiguous()
        self._chk(self.lib.sphx_dev_species(self.ctx.h, S, self._p(fun), self._p(m), self._p(F), self._p(Z), self._p(A)))
        return F, Z, A

    # ---- the loop forms of the reference's time loop (drv:451-458) on owned + ghost arrays ----
    def loop_prep(self, pos, vel, m, T, mu, gam, ptype, E, d):
        self._keep = (pos, vel, m, T, mu, gam, ptype, E)
        self._chk(self.lib.sphx_dev_loop_prep(self.ctx.h, *[self._p(t) for t in self._keep], float(d)))

    def loop_pass1(self, h_complete):
        n, dev = self.n_total, self.device
        rho = torch.empty(n, dtype=torch.float64, device=dev)
        rhod = torch.empty(n, dtype=torch.float64, device=dev)
        nden = torch.empty(n, dtype=torch.float64, device=dev)
        delp = torch.empty((n, 3), dtype=torch.float64, device=dev)
        self._chk(self.lib.sphx_dev_loop_pass1(self.ctx.h, self._p(h_complete), self._p(rho), self._p(rhod), self._p(nden),
                                               self._p(delp)))
        return rho, rhod, nden, delp

    def loop_pass2_interior(self):
        """Pass 2 of the blobs without ghost neighbours, to be called while rho_j travels (sphx_dev_loop_pass2_interior);
        True if launched - loop_pass2 then finishes with the boundary blobs."""
        rc = self.lib.sphx_dev_loop_pass2_interior(self.ctx.h)
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def blob_split_counts(self):
        """[interior, boundary, idle] blobs of the last search (a host read: diagnostics only)."""
        out = (C.c_int32 * 3)()
        self._chk(self.lib.sphx_dev_blob_split_counts(self.ctx.h, out))
        return [int(v) for v in out]

    def loop_pass2(self, rho_complete):
        n, dev = self.n_total, self.device
        va = torch.empty((n, 3), dtype=torch.float64, device=dev)
        vh = torch.empty(n, dtype=torch.float64, device=dev)
        ct = torch.zeros(1, dtype=torch.float64, device=dev)
        self._chk(self.lib.sphx_dev_loop_pass2(self.ctx.h, self._p(rho_complete), self._p(va), self._p(vh), self._p(ct)))
        return va, vh, ct

    def integrate_loop(self, n_owned, pos, vel, acc, E, T, m, mu, gam, ptype, delp, rho, va, vh, dt, red2=None,
                       first=False, fixed_dt=0.0):
        """drv:460-491 on the loop forms' outputs; red2 given: verdict and dt on the device -> dt as a (1,) tensor."""
        dt_t = torch.empty(1, dtype=torch.float64, device=self.device) if red2 is not None else None
        self._chk(self.lib.sphx_dev_integrate_loop(self.ctx.h, n_owned, self._p(pos), self._p(vel), self._p(acc),
                                                   self._p(E), self._p(T), self._p(m), self._p(mu), self._p(gam),
                                                   self._p(ptype), self._p(delp), self._p(rho), self._p(va), self._p(vh),
                                                   self._p(red2), 1 if first else 0, float(fixed_dt), float(dt),
                                                   self._p(dt_t)))
        return dt_t

    def integrate(self, n_owned, pos, vel, acc, E, T, m, mu, gam, ptype, ha, va, vh, dt):
        self._chk(self.lib.sphx_dev_integrate(self.ctx.h, n_owned, self._p(pos), self._p(vel), self._p(acc),
                                              self._p(E), self._p(T), self._p(m), self._p(mu), self._p(gam),
                                              self._p(ptype), self._p(ha), self._p(va), self._p(vh), float(dt)))


# ==============================================================================================
# the decomposed step loop
# ==============================================================================================
GHOST_FIELDS = ("vel", "m", "T", "mu", "gam", "ptype", "h")      # + pos; sent in halo phase 1
OWNED_FIELDS = ("pos", "vel", "acc", "m", "T", "mu", "gam", "ptype", "E", "h", "gid")


class DistributedSim:
    """Owned particles of one rank + the per-step halo protocol.  `state` holds this rank's
    owned particles (reference array conventions, sph/code_running.py:114-177) plus 'gid'
    (global particle ids); `lo`/`hi` are the (world,3) region boxes of rcb_regions."""

    phase = "setup"                              # the section of step() this process is in (class-wide: one sim per process)
    DT_0 = 60. * 60. * 24. * 365. * 250000.      # nsc:38
    MAX_AGE = 3e7 * 60. * 60. * 24. * 365.       # drv:79

    def __init__(self, state, lo, hi, backend, rank=0, world=1, device="cpu", comm_device=None,
                 halo_scale=1.15, skin_frac=0.15, need_grid=96, migrate_every=4, forms="hydro_update", d=None,
                 with_drag=False, with_species=False, agb=None):
        """forms: the sums of the step - "hydro_update" (nsc:556-671; four halo phases) or "loop", the loop forms the
        reference's time loop calls (drv:451-458, smoothing length from the driver's global `d`, drv:68; three halo
        phases: state + E, h_j, rho_j - no Pi_j)."""
        if forms not in ("hydro_update", "loop"):
            raise ValueError("forms must be 'hydro_update' or 'loop'")
        if forms == "loop" and not (d is not None and d > 0):
            raise ValueError("forms='loop' needs the driver's global d (code_running.py:67-68)")
        self.forms, self.d = forms, (float(d) if d is not None else None)
        # with_drag: the dust -> gas drag of nsc.net_impulse in the step (drv:455,462-463,473): ghosts carry their mean
        # grain mass / cross-section, and the reaction scatter-added onto ghost neighbours travels back to their owners
        # (the reverse halo).  with_species: every step forms f_un_neighbor (nsc:624-627) for the owned particles (ghosts
        # carry their composition rows); agb = (splines, mapto, divisor): also the metallicity and the AGB yields.
        self.with_drag, self.with_species, self.agb = bool(with_drag), bool(with_species or agb is not None), agb
        drag_given = state.get("mean_grain_mass") is not None and state.get("mean_cross_section") is not None
        if (self.with_species or (self.with_drag and not drag_given)) and state.get("f_un") is None:
            raise ValueError("with_drag / with_species need a state with f_un")
        self.rank, self.world, self.backend = rank, world, backend
        self.device = torch.device(device)
        self.comm_device = torch.device(comm_device) if comm_device is not None else self.device
        self.ex = Exchanger(rank, world, self.comm_device)
        self.ex.warm_up()
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)
        self.s = dict(pos=f(state["points"]), vel=f(state["velocities"]), m=f(state["mass"]), T=f(state["T"]),
                      mu=f(state["mu_array"]), gam=f(state["gamma_array"]), ptype=f(state["particle_type"]),
                      E=f(state["E_internal"]))
        n = self.s["pos"].shape[0]
        acc = state.get("total_accel")
        self.s["acc"] = f(acc) if acc is not None else torch.zeros((n, 3), dtype=torch.float64, device=self.device)
        self.s["h"] = torch.zeros(n, dtype=torch.float64, device=self.device)
        self.s["gid"] = torch.as_tensor(np.asarray(state["gid"], dtype=np.int64)).to(self.device)
        self.extra_fields = []                  # per-particle fields beyond the core set: migrate and travel as ghosts
        if self.with_drag:
            if drag_given:                      # (a snapshot: the per-particle means as the run carried them)
                self.s["mgm"], self.s["mcs"] = f(state["mean_grain_mass"]), f(state["mean_cross_section"])
            else:
                from . import compat
                fu = np.asarray(state["f_un"], dtype=np.float64)
                self.s["mgm"] = f(np.sum(compat.grain_mass() * fu, axis=1))          # nsc:725
                self.s["mcs"] = f(np.sum(compat.sigma_effective() * fu, axis=1))     # nsc:726
            self.extra_fields += ["mgm", "mcs"]
        if self.with_species:
            self.s["fun"] = f(state["f_un"])
            self.extra_fields += ["fun"]
            if agb is not None and hasattr(backend, "set_agb"):
                from . import compat
                backend.set_agb(int(self.s["fun"].shape[1]), agb, compat.mu_specie, compat.solar_mass)
        self.lo = torch.as_tensor(lo, dtype=torch.float64).to(self.device)
        self.hi = torch.as_tensor(hi, dtype=torch.float64).to(self.device)
        self.halo_scale = halo_scale
        self.skin_frac = skin_frac          # plan is reused while displacements stay below skin/2
        self.w_plan, self.send_idx, self.recv_counts, self.pos_plan = None, None, None, None
        self.send_cat = None
        self.grow = 1.0
        # strays are handed to their region's owner (and the owned arrays re-sorted) on every
        # `migrate_every`-th replan: ownership is bookkeeping, the halo is built from where the owned
        # particles ARE, so a particle a little outside its region's box costs nothing but reach
        self.migrate_every = max(1, int(migrate_every))
        # when the plan went stale on three steps running (large dt: some particle always moves more
        # than the skin), the next steps replan without asking - two host-synchronising reductions less
        self.stale_streak, self.force_replan = 0, 0
        self._nonzero_static = None       # torch.nonzero_static usable on this device? (found out at the first plan)
        # how much of |v_i| dt a particle adds to the reach it claims (1: a radius may grow by the particle's own displacement
        # per step; every plan is verified after the search and redone if too thin, so this only trades ghosts for redos)
        self.reach_vfac = float(os.environ.get("SPHX_REACH_VFAC", "1.0"))
        # head-room of a claim limited to this many MEAN radii (0: not): the rim of an expanding cloud (radii tens of times
        # the mean, growing by a fraction of a per cent per step) otherwise claims 30 % of its own radius into the dense core
        self.reach_cap = float(os.environ.get("SPHX_REACH_CAP", "1.0"))
        self.cap_abs = 0.0                # the cap as a length: set when particles migrate (the same for every plan in between)
        self.overlap = os.environ.get("SPHX_MG_OVERLAP", "1") != "0"      # interior blobs' sums under the scalar halo phases
        self.plan_next = None             # the NEXT step's plan, made at the end of this one (see step)
        self.plan_ahead = True
        # coarse global grid for the need maps: global bounding box of the initial state + 25 %
        pmin = self.s["pos"].min(dim=0).values if n else torch.full((3,), 1e300, dtype=torch.float64, device=self.device)
        pmax = self.s["pos"].max(dim=0).values if n else torch.full((3,), -1e300, dtype=torch.float64, device=self.device)
        if world > 1:
            t0 = pmin.to(self.comm_device); t1 = pmax.to(self.comm_device)
            dist.all_reduce(t0, op=dist.ReduceOp.MIN); dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            pmin, pmax = t0.to(self.device), t1.to(self.device)
        ext = torch.clamp(pmax - pmin, min=1e-300)
        self.G = int(need_grid)
        self.g_lo = pmin - 0.25 * ext
        self.g_cs = float((1.5 * ext).max()) / self.G             # cubic coarse cells
        self.g_lo_host = [float(v) for v in self.g_lo.tolist()]
        self.hmax_prev, self.hmean_prev = 0.0, 0.0
        self.last_ntotal = None
        self.first = True
        self.dt_last = 0.0
        self.last = {}
        self.stats = dict(steps=0, ghosts=0, redo=0, migrated=0)
        self.host_ms = {}

    # ------------------------------------------------------------------------------------------
    class _Section:
        """Host wall time of a section of step(), accumulated in sim.host_ms[name].  No extra
        synchronisation: GPU work is charged to the section whose host code next waits for it."""
        __slots__ = ("acc", "name", "t")

        def __init__(self, acc, name):
            self.acc, self.name = acc, name

        def __enter__(self):
            DistributedSim.phase = self.name             # (what a failing rank names when it gives up: bench_main)
            self.t = time.perf_counter()

        def __exit__(self, *exc):
            self.acc[self.name] = self.acc.get(self.name, 0.0) + (time.perf_counter() - self.t) * 1e3
            return False

    def _sec(self, name):
        return self._Section(self.host_ms, name)

    @property
    def n_owned(self):
        return self.s["pos"].shape[0]

    def _allgather_scalar(self, v):
        if self.world == 1:
            return [float(v)]
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.comm_device)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [float(o[0]) for o in out]

    def _allreduce_min(self, v):
        if self.world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t[0])

    def _allreduce_max(self, v):
        if self.world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def _coarse_cell(self, pos):
        c = torch.floor((pos - self.g_lo[None]) / self.g_cs).to(torch.int64).clamp_(0, self.G - 1)
        return (c[:, 2] * self.G + c[:, 1]) * self.G + c[:, 0]

    def _need_map(self, w_owned):
        """uint8 (G^3,): coarse cells from which a particle could lie within w_i of one of my owned particles.  A cell
        d = (dx, dy, dz) cells away from an owned particle's cell is marked when sum_a max(|d_a| - 1, 0)^2 <= (w_i / cell)^2:
        a sphere of cells (distances under-estimated by one cell per axis), not the cube around it - the wide claims of
        rim particles would otherwise pull in the dense cloud at the cube's corners."""
        G, cs = self.G, self.g_cs
        # (a NaN reach - a broken radius - claims everything, in both forms: as a NaN it would drop out of the kernel's
        #  comparisons and, in the form below, swallow its cell's genuine claims in the running maximum)
        if hasattr(self.backend, "need_map"):          # fused: one launch, no host synchronisation
            return self.backend.need_map(self.s["pos"], w_owned, self.g_lo_host, cs, G)
        # the tensor-library form (the specification: what the CPU tests run, what the kernel is tested against).  The
        # rule is monotone in w, so a cell's widest claim stands for all its particles; cells are handled per integer
        # radius class, each against its stencil of offsets.
        dev = self.device
        out = torch.zeros(G * G * G, dtype=torch.uint8, device=dev)
        if not w_owned.numel():
            return out
        w_owned = torch.nan_to_num(w_owned, nan=float("inf"), posinf=float("inf"), neginf=0.0)
        wmax = torch.zeros(G * G * G, dtype=torch.float64, device=dev)
        wmax.scatter_reduce_(0, self._coarse_cell(self.s["pos"]), w_owned, reduce="amax", include_self=True)
        occ = torch.nonzero(wmax > 0).flatten()
        if not occ.numel():
            return out
        q = wmax[occ] / cs
        r = torch.clamp(torch.floor(q) + 1.0, max=float(G)).to(torch.int64)
        q2 = q * q
        cz, cy, cx = occ // (G * G), (occ // G) % G, occ % G
        for rv in torch.unique(r).tolist():
            sel = torch.nonzero(r == rv).flatten()
            lo_r = max(-rv, -(G - 1)); hi_r = min(rv, G - 1)
            o1 = torch.arange(lo_r, hi_r + 1, device=dev, dtype=torch.int64)
            a1 = torch.clamp(o1.abs() - 1, min=0)
            # stencil of this radius class: offsets and their under-estimated squared distance in cells
            dz, dy, dx = torch.meshgrid(o1, o1, o1, indexing="ij")
            s2 = (a1 * a1)[:, None, None] + (a1 * a1)[None, :, None] + (a1 * a1)[None, None, :]
            dz, dy, dx, s2 = dz.reshape(-1), dy.reshape(-1), dx.reshape(-1), s2.reshape(-1).to(torch.float64)
            chunk = max(1, int(4_000_000 // max(int(s2.numel()), 1)))
            for c0 in range(0, int(sel.numel()), chunk):
                ss = sel[c0:c0 + chunk]
                hit = s2[None, :] <= q2[ss][:, None]                                  # (cells, offsets)
                z = cz[ss][:, None] + dz[None, :]; y = cy[ss][:, None] + dy[None, :]; x = cx[ss][:, None] + dx[None, :]
                hit &= (z >= 0) & (z < G) & (y >= 0) & (y < G) & (x >= 0) & (x < G)
                out[((z * G + y) * G + x)[hit]] = 1
        return out

    def _plan_launch(self, w_owned):
        """Device work and collectives of a plan (one all_gather of G^3 bytes per rank, one of W counts): -> (mask (W, n),
        both (2, W) int64 on the communication device: what I send to / receive from every peer).  No host read."""
        W = self.world
        mine = self._need_map(w_owned).to(self.comm_device)
        maps = [torch.zeros_like(mine) for _ in range(W)]
        dist.all_gather(maps, mine)
        maps = torch.stack(maps).to(self.device)                              # (W, G^3)
        if hasattr(self.backend, "plan_mask"):
            mask, cnt = self.backend.plan_mask(self.s["pos"], self.g_lo_host, self.g_cs, self.G, maps, self.rank)
        else:
            cell = self._coarse_cell(self.s["pos"])
            mask = maps[:, cell] != 0                                         # (W, n)
            mask[self.rank] = False
            cnt = mask.sum(dim=1)
        # counts to every peer, exchanged while still on the device: ONE host read (the caller's) serves both the
        # send lists and what the peers will send (each read drains the stream)
        cnt = cnt.to(self.comm_device)                                        # (W,) int64
        allc = [torch.zeros_like(cnt) for _ in range(W)]
        dist.all_gather(allc, cnt)
        return mask, torch.stack([cnt, torch.stack(allc)[:, self.rank]])

    def _plan_finish(self, mask, both):
        """Send lists from the mask, now that the host knows the counts (`both` as nested lists)."""
        counts, recv_counts = [int(v) for v in both[0]], [int(v) for v in both[1]]
        total = sum(counts)
        pk = None
        if self._nonzero_static is not False:                                 # sized by the counts: no second host read
            try:
                pk = torch.nonzero_static(mask, size=total)
                self._nonzero_static = True
            except (RuntimeError, NotImplementedError, AttributeError):
                if self._nonzero_static:
                    raise
                self._nonzero_static = False
        if pk is None:
            pk = torch.nonzero(mask)                                          # sorted by peer, then particle
        send_idx, o = [], 0
        for p in range(self.world):
            send_idx.append(None if p == self.rank else pk[o:o + counts[p], 1].contiguous())
            o += counts[p]
        return send_idx, recv_counts

    def _plan(self, w_owned):
        """Send lists from the ranks' need maps."""
        mask, both = self._plan_launch(w_owned)
        return self._plan_finish(mask, both.tolist())

    # rows <-> separate arrays.  Backends with fused kernels (LibBackend) do each in one launch; the
    # tensor-library forms below are the specification (and what the CPU tests run).
    @staticmethod
    def _as_f64(t):
        return t.view(torch.float64) if t.dtype == torch.int64 else t        # ids travel as raw 8-byte words

    def _pack(self, idx, fields):
        """(len(idx), W) rows of the particles idx (None: all) from the given (n,) / (n,w) fields."""
        fields = [self._as_f64(f) for f in fields]
        if hasattr(self.backend, "pack_rows"):
            return self.backend.pack_rows(idx, fields)
        rows = torch.cat([f if f.dim() == 2 else f[:, None] for f in fields], dim=1)
        return rows if idx is None else rows[idx]

    def _regroup(self, sel, n_sel, rows, fields):
        """fields[sel] (None: the first n_sel as they are) followed by the particles in rows."""
        dts = [f.dtype for f in fields]
        f64 = [self._as_f64(f) for f in fields]
        if hasattr(self.backend, "regroup"):
            outs = self.backend.regroup(sel, n_sel, rows, f64)
        else:
            outs, c = [], 0
            for f in f64:
                w = 1 if f.dim() == 1 else f.shape[1]
                own = f[:n_sel] if sel is None else f[sel]
                if rows is not None and rows.shape[0]:
                    g = rows[:, c] if f.dim() == 1 else rows[:, c:c + w]
                    own = torch.cat([own, g], dim=0)
                outs.append(own.contiguous())
                c += w
        return [o.view(torch.int64) if d == torch.int64 else o for o, d in zip(outs, dts)]

    def _exchange(self, send_idx, recv_counts, fields, into=None, send_cat=None, defer=False):
        """Rows of `fields` for every peer's send list -> the received rows (n_recv, W), ordered by
        source rank.  `into`: receive in place (a contiguous (n_recv, W) view on the compute device).
        defer=True: started only; the returned function completes the exchange and gives the rows."""
        W = sum(1 if f.dim() == 1 else f.shape[1] for f in fields)
        dev = fields[0].device
        nrecv = sum(recv_counts)
        if self.world == 1:
            res1 = into if into is not None else torch.zeros((0, W), dtype=torch.float64, device=dev)
            return (lambda: res1) if defer else res1
        if send_cat is None:
            parts = [ix for ix in send_idx if ix is not None and ix.numel()]
            send_cat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
        rows = self._pack(send_cat, fields)
        bufs, o = [], 0
        for ix in send_idx:
            c = 0 if ix is None else int(ix.numel())
            bufs.append(None if ix is None else rows[o:o + c])
            o += c
        direct = into is not None and into.device == self.comm_device and into.is_contiguous()
        if into is None and dev == self.comm_device:
            into = torch.empty((nrecv, W), dtype=torch.float64, device=dev)
            direct = True
        pending = self.ex.rows(bufs, recv_counts, W, into=into if direct else None, defer=True)

        def finish():
            got = pending()
            if direct:
                return into
            parts = [g for g in got if g is not None]
            res = torch.cat(parts, dim=0) if parts else torch.zeros((0, W), dtype=torch.float64, device=dev)
            if into is not None:
                into.copy_(res)
                return into
            return res
        return finish if defer else finish()

    # ------------------------------------------------------------------------------------------
    def _migration_due(self):
        k_ = self.stats.get("replans", 0)
        return k_ < 2 or k_ % self.migrate_every == 0       # (the first two: one-off set-up costs paid early)

    def _replan(self):
        """Migrate strays to their region's owner, then rebuild the send lists: every owned particle
        claims the radius w_i = (halo_scale + skin_frac) * grow * h_i around itself."""
        if self._migration_due():
            self._reorder_and_migrate()
            self.cap_abs = self.reach_cap * self.hmean_prev if self.hmean_prev > 0.0 else 0.0
            if hasattr(self.backend, "set_reach_cap"):
                self.backend.set_reach_cap(self.cap_abs)
        s = self.s
        # reach claimed by each owned particle: its kNN radius with head-room for one step of growth
        # (a radius changes by at most twice the local displacement, so fast movers claim more)
        if hasattr(self.backend, "reach") and self.n_owned:
            self.w_plan = self.backend.reach(s["h"], s["vel"], self.halo_scale, self.skin_frac, self.dt_last * self.reach_vfac)
        else:
            speed = torch.sqrt((s["vel"] * s["vel"]).sum(dim=1))
            dtv = self.dt_last * self.reach_vfac
            if self.cap_abs > 0.0:
                cap = torch.full_like(s["h"], self.cap_abs)
                a = torch.minimum((self.halo_scale + self.skin_frac - 1.0) * s["h"], cap)
                b = torch.minimum((self.halo_scale - 1.0) * s["h"], cap)
                self.w_plan = torch.maximum(s["h"] + a, (s["h"] + b) + speed * dtv)
            else:
                self.w_plan = torch.maximum((self.halo_scale + self.skin_frac) * s["h"], self.halo_scale * s["h"] + speed * dtv)
        if self.world > 1:
            self.send_idx, self.recv_counts = self._plan(self.w_plan)
        else:
            self.send_idx, self.recv_counts = [None], [0]
        parts = [ix for ix in self.send_idx if ix is not None and ix.numel()]
        self.send_cat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=self.device)
        # (positions at plan time: only read by the staleness check, which forced replanning skips)
        self.pos_plan = s["pos"].clone() if self.force_replan <= 0 else None
        self.stats["replans"] = self.stats.get("replans", 0) + 1

    def _cell_order(self):
        """Owned particles in the cell order of the last search (when the backend exposes it), or None.
        Keeping the owned arrays in that order makes the library's gathers / scatters between caller
        order and cell order nearly sequential, as in the fused single-GPU loop."""
        if self.last_ntotal is None or not hasattr(self.backend, "order"):
            return None
        no = self.n_owned
        order = self.backend.order(self.last_ntotal)
        order = order[order < no]
        self.last_ntotal = None
        return order if order.numel() == no else None

    def _reverse_add(self, send_idx, recv_counts, ghost_rows, owned_acc):
        """The reverse halo: rows accumulated on my ghosts go back to their owners, who add them to the particles they
        sent (a particle sent to several peers collects from each; fixed order of addition)."""
        if self.world == 1:
            return
        W = int(ghost_rows.shape[1])
        bufs, o = [], 0
        for p in range(self.world):
            c = 0 if p == self.rank else int(recv_counts[p])
            bufs.append(None if p == self.rank else ghost_rows[o:o + c])
            o += c
        back = [0 if ix is None else int(ix.numel()) for ix in send_idx]
        got = self.ex.rows(bufs, back, W)
        for p in range(self.world):
            if got[p] is not None and back[p]:
                owned_acc.index_add_(0, send_idx[p], got[p].to(owned_acc.device))

    def _step_extras(self, be, no, ng, send_idx, recv_counts, m, ptype, extra, rho, rhod):
        """Drag (nsc:719-742; the completed terms are handed to the coming update) and the species pass (nsc:624-627)."""
        if self.with_drag:
            onto, react = be.drag(m, ptype, extra["mgm"], extra["mcs"])
            own = react[:no].clone()
            self._reverse_add(send_idx, recv_counts, react[no:].contiguous(), own)
            be.set_drag_terms(onto[:no], own, rho[:no], rhod[:no])
        if self.with_species:
            F, Z, A = be.species(extra["fun"], m)
            self.last_species = dict(f_un_neighbor=F[:, :no], metallicity=None if Z is None else Z[:no],
                                     agb_dust=None if A is None else A[:no])

    def step(self, fixed_dt=0.0):
        """One decomposed pass of the hot path.

        Halo validity.  When the plan was made every foreign particle within w_i of owned particle
        i became a ghost.  With D the largest displacement of ANY particle since, a foreign
        non-ghost is still farther than w_i - 2D from i, so the neighbours within h_i of i are all
        owned or ghosts as long as  h_i + 2D <= w_i.  That is checked for every owned particle after
        every search; if it fails anywhere all ranks replan (with a larger factor) and redo the
        search.  A plan is reused as long as halo_scale*h_i + 2D <= w_i holds everywhere."""
        s, be = self.s, self.backend
        be.clamp(s["pos"], s["vel"])                                        # drv:233-238
        no = self.n_owned
        bootstrap = self.first or self.w_plan is None or self.hmax_prev <= 0.0
        if bootstrap:
            # a search without ghosts over-estimates every h (fewer candidates): a rigorous radius
            # for the first exchange
            if no:
                s["h"] = be.search(s["pos"], no, None, 0.0)[:no].contiguous()
            with self._sec("replan"):
                self._replan()
            D = 0.0
        elif self.force_replan > 0 or self.pos_plan is None:
            # forced, or the last plan kept no positions to measure the displacement against: a plan
            # made now is exact (D = 0); never compare against positions the plan was not made at
            self.force_replan = max(self.force_replan - 1, 0)
            with self._sec("replan"):
                if self.plan_next is not None:
                    # made at the end of the previous step on the very positions this step starts from (its
                    # collectives behind that step's sums, its counts in that step's one host read)
                    self.w_plan, self.send_idx, self.recv_counts, self.send_cat = self.plan_next
                    self.pos_plan = None
                    self.stats["replans"] = self.stats.get("replans", 0) + 1
                else:
                    self._replan()
            self.plan_next = None
            D = 0.0
        else:
            with self._sec("stale_check"):
                if no:
                    d_loc = float((s["pos"] - self.pos_plan).abs().max()) * 1.7320508075688772
                else:
                    d_loc = 0.0
                D = self._allreduce_max(d_loc)                    # bound on the displacement norm
                hr = (self.halo_scale - 1.0) * s["h"]              # (the head-room a plan is expected to keep)
                if self.cap_abs > 0.0:
                    hr = torch.clamp(hr, max=self.cap_abs)
                stale = float((s["h"] + hr + 2.0 * D > self.w_plan).any()) if no else 0.0
                is_stale = self._allreduce_max(stale) > 0.5       # (the same verdict on every rank)
            if is_stale:
                with self._sec("replan"):
                    self._replan()
                D = 0.0
                self.stale_streak += 1
                if self.stale_streak >= 3:
                    self.stale_streak, self.force_replan = 0, 16
            else:
                self.stale_streak = 0
        while True:
            s, no = self.s, self.n_owned                      # (a replan may have migrated particles)
            send_idx, recv_counts = self.send_idx, self.recv_counts
            ng = sum(recv_counts)
            hint_owned = s["h"]
            mean_h = self.hmean_prev
            # ---- halo phase 1: ghosts' state (88 B + previous h) ------------------------------
            loop = self.forms == "loop"
            state_keys = ["pos", "vel", "m", "T", "mu", "gam", "ptype", "h"]
            if loop:
                state_keys.append("E")                    # del_pressure reads the neighbour's E (nsc:755)
            state_keys += list(self.extra_fields)         # drag coefficients / composition rows of the ghosts
            state_fields = [s[k_] for k_ in state_keys]
            with self._sec("halo_state"):
                g = self._exchange(send_idx, recv_counts, state_fields, send_cat=self.send_cat)
                if hasattr(be, "append_rows") and all(f.dtype == torch.float64 for f in state_fields):
                    # the ghosts go behind the owned particles in the owned arrays' own storage: nothing else is copied
                    regrouped, owned = be.append_rows(no, g, state_fields)
                    for k_, v in zip(state_keys, owned):
                        s[k_] = v
                else:
                    regrouped = self._regroup(None, no, g, state_fields)
                pos, vel, m, T, mu, gam, ptype, hint = regrouped[:8]
                E_all = regrouped[8] if loop else None
                extra = dict(zip(self.extra_fields, regrouped[(9 if loop else 8):]))
            with self._sec("search"):
                h = be.search(pos, no, hint, mean_h)
            f64 = dict(dtype=torch.float64, device=h.device)
            fused = hasattr(be, "step_scalars") and no > 0
            # a kNN radius that outgrew its claimed reach (checked below, together with dt: the sums
            # are run on the assumption that the halo was sufficient, which it nearly always is)
            if not fused:
                bad_t = (h[:no] + 2.0 * D > self.w_plan).any().to(torch.float64).reshape(1) if no else torch.zeros(1, **f64)
            # ---- halo phases 2-4: 8 B per ghost, received straight into the tail of the (n_total,)
            # array the library just filled for the owned particles ----------------------------
            tail = lambda a: a[no:].view(ng, 1)
            if loop:
                with self._sec("sums+halo_scalars"):
                    # h_j (nsc:711) travels while the step's records are built (they do not read h)
                    h_done = self._exchange(send_idx, recv_counts, [h[:no]], into=tail(h), send_cat=self.send_cat, defer=True)
                    be.loop_prep(pos, vel, m, T, mu, gam, ptype, E_all, self.d)
                    h_done()
                    rho, rhod, nden, delp = be.loop_pass1(h)
                    # rho_j (nsc:803) travels while pass 2 runs for the blobs without ghost neighbours
                    rho_done = self._exchange(send_idx, recv_counts, [rho[:no]], into=tail(rho), send_cat=self.send_cat,
                                              defer=True)
                    if self.overlap and hasattr(be, "loop_pass2_interior"):
                        be.loop_pass2_interior()
                    rho_done()
                    va, vh, ct = be.loop_pass2(rho)
                    ha = None
                    if self.with_drag or self.with_species:      # both read hydro_update's records (complete h)
                        be.prep(pos, vel, m, h, T, mu, gam, ptype)
                    self._step_extras(be, no, ng, send_idx, recv_counts, m, ptype, extra, rho, rhod)
            else:
              with self._sec("sums+halo_scalars"):
                xch = lambda a, defer=False: self._exchange(send_idx, recv_counts, [a[:no]], into=tail(a), send_cat=self.send_cat,
                                                            defer=defer)
                if self.overlap and self.world > 1 and hasattr(be, "select_blobs") and be.select_blobs(1):
                    # every scalar phase travels while the pass that follows it runs for the blobs whose particles have
                    # no ghost neighbour (they read nothing the phase delivers); the boundary blobs run after it
                    done = xch(h, True)                                        # h_j
                    be.prep(pos, vel, m, h, T, mu, gam, ptype)                 # (owned particles' records)
                    o1 = be.density(self.with_drag)
                    done(); be.select_blobs(2)
                    be.prep(pos, vel, m, h, T, mu, gam, ptype)                 # (ghosts' records)
                    rho, nden, ha = be.density(self.with_drag, out=o1)
                    done = xch(rho, True); be.select_blobs(1)                  # rho_j
                    o2 = be.pi(rho)
                    done(); be.select_blobs(2)
                    bw, ct = be.pi(rho, out=o2)
                    done = xch(bw, True); be.select_blobs(1)                   # m Pi_j
                    o3 = be.visc(bw, m)
                    done(); be.select_blobs(2)
                    va, vh = be.visc(bw, m, out=o3)
                    be.select_blobs(0)
                else:
                    xch(h)                                                     # h_j
                    be.prep(pos, vel, m, h, T, mu, gam, ptype)
                    rho, nden, ha = be.density(True) if self.with_drag else be.density()
                    xch(rho)                                                   # rho_j
                    bw, ct = be.pi(rho)
                    xch(bw)                                                    # m Pi_j
                    va, vh = be.visc(bw, m)
                self._step_extras(be, no, ng, send_idx, recv_counts, m, ptype, extra, rho,
                                  getattr(be, "rhod", None) if self.with_drag else None)
            # ---- ONE reduction and ONE host read for the step's scalars: halo verdict (max), global
            # minimum crossing time for dt (nsc:786, drv:222-229; as max of the negative), max / mean h
            t_dt = time.perf_counter()
            ctt = ct.reshape(-1)[:1].to(**f64) if torch.is_tensor(ct) else torch.tensor([float(ct)], **f64)
            if fused:                              # one launch: {verdict, -ct, max h, robust mean h}
                out4 = be.step_scalars(no, h, self.w_plan, D, 8.0 * self.hmean_prev if self.hmean_prev > 0.0 else 0.0,
                                       ctt)
                red, loc = out4[:2], out4[2:]
            else:
                red = torch.cat([bad_t, -ctt])
                if no:
                    ho = h[:no]
                    if self.hmean_prev > 0.0:          # escapers' radii must not size the cells (robust mean)
                        keep = ho <= 8.0 * self.hmean_prev
                        hm = (ho * keep).sum() / keep.sum().clamp(min=1)
                    else:
                        hm = ho.mean()
                    loc = torch.stack([ho.max(), hm])
                else:
                    loc = torch.zeros(2, **f64)
            if self.world > 1:
                if red.device == self.comm_device:
                    dist.all_reduce(red, op=dist.ReduceOp.MAX)          # (in place: a view of out4 when fused)
                else:
                    red = red.to(self.comm_device)
                    dist.all_reduce(red, op=dist.ReduceOp.MAX)
                    red = red.to(h.device)
            auto = fused and hasattr(be, "integrate_auto")
            # (the same on every rank: the plan's collectives are entered by all or by none)
            ahead_ok = (self.plan_ahead and self.world > 1 and self.force_replan > 0 and not self._migration_due()
                        and all(hasattr(be, a_) for a_ in ("reach_dt", "plan_mask", "step_scalars", "integrate_auto")))
            ahead = None
            if ahead_ok and not auto:
                # a rank without owned particles: nothing to send, but the collectives are collective
                with self._sec("plan_ahead"):
                    mask_next, both_next = self._plan_launch(torch.zeros(0, **f64))
                    ahead = (torch.zeros(0, **f64), mask_next)
                    both_host = both_next.tolist()
            if auto:
                # the update is launched before the host learns the verdict: the kernel itself leaves the state
                # alone when the step has to be redone, and works out dt from the reduced crossing time
                red_dev = red if red.device == h.device else red.to(h.device)
                if loop:
                    dt_t = be.integrate_loop(no, s["pos"], s["vel"], s["acc"], s["E"], s["T"], s["m"], s["mu"], s["gam"],
                                             s["ptype"], delp, rho, va, vh, 0.0, red2=red_dev.contiguous(),
                                             first=self.first, fixed_dt=fixed_dt)
                else:
                    dt_t = be.integrate_auto(no, s["pos"], s["vel"], s["acc"], s["E"], s["T"], s["m"], s["mu"], s["gam"],
                                             s["ptype"], ha, va, vh, red_dev.contiguous(), self.first, fixed_dt)
                # ---- the NEXT step's plan, while the host still has not looked at this one's scalars: the update is
                # on its way, so the positions the next step starts from are known to the device.  Reach, need map,
                # both all_gathers and the send mask are launched now; their counts ride in this step's one host read.
                # (Speculative: a "redo" verdict leaves the state untouched and this plan is dropped.)
                if ahead_ok:
                    with self._sec("plan_ahead"):
                        be.clamp(s["pos"], s["vel"])            # drv:233-238, early: the next step's own clamp finds nothing to do
                        w_next = be.reach_dt(h[:no].contiguous(), s["vel"], self.halo_scale, self.skin_frac,
                                             dt_t if self.reach_vfac == 1.0 else dt_t * self.reach_vfac)
                        mask_next, both_next = self._plan_launch(w_next)
                        ahead = (w_next, mask_next)
                if ahead is not None and both_next.device == red_dev.device:
                    vals = torch.cat([red_dev, loc, dt_t, both_next.reshape(-1).to(torch.float64)]).tolist()
                    both_host = [vals[5:5 + self.world], vals[5 + self.world:5 + 2 * self.world]]
                else:
                    vals = torch.cat([red_dev, loc, dt_t]).tolist()
                    both_host = both_next.tolist() if ahead is not None else None     # (host-memory halo: already there)
            else:
                vals = out4.tolist() if (fused and red.data_ptr() == out4.data_ptr()) else torch.cat([red, loc]).tolist()
            if vals[0] < 0.5:
                break
            # the halo was too thin somewhere: those particles claim their new radius (x1.5), everybody
            # replans and the step's search and sums are redone
            if no:
                fail = h[:no] + 2.0 * D > self.w_plan
                s["h"] = torch.where(fail, 1.5 * (h[:no] + 2.0 * D), s["h"])
            # (a migration inside the redo sorts by THIS attempt's search: the state has not moved since)
            self.last_ntotal = int(pos.shape[0])
            with self._sec("replan"):
                self._replan()
            D = 0.0
            self.stats["redo"] += 1
        ct_min, self.hmax_prev, self.hmean_prev = -vals[1], vals[2], vals[3]
        self.plan_next = None
        if ahead is not None:
            with self._sec("plan_ahead"):
                send_idx_n, recv_n = self._plan_finish(ahead[1], both_host)
                parts = [ix for ix in send_idx_n if ix is not None and ix.numel()]
                cat_n = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=self.device)
                self.plan_next = (ahead[0], send_idx_n, recv_n, cat_n)
        if auto:
            dt = vals[4]                                   # (worked out and applied on the device)
        else:
            ctv = self.DT_0 / 10. if ct_min >= HUGE_CT else ct_min + 0.0001
            if fixed_dt > 0:
                dt = fixed_dt
            else:
                dt = self.DT_0 / 10. if self.first else max(self.DT_0 / 5., min(self.DT_0 * 2., ctv))
                if ctv > self.MAX_AGE:
                    dt = self.MAX_AGE / 100.
            if self.forms == "loop":
                be.integrate_loop(no, s["pos"], s["vel"], s["acc"], s["E"], s["T"], s["m"], s["mu"], s["gam"], s["ptype"],
                                  delp, rho, va, vh, dt)
            else:
                be.integrate(no, s["pos"], s["vel"], s["acc"], s["E"], s["T"], s["m"], s["mu"], s["gam"], s["ptype"],
                             ha, va, vh, dt)
        self.host_ms["dt+integrate"] = self.host_ms.get("dt+integrate", 0.0) + (time.perf_counter() - t_dt) * 1e3
        s["h"] = h[:no].contiguous()
        self.last = dict(rho=rho[:no], nden=nden[:no], visc_heat=vh[:no])
        self.dt_last = dt
        self.last_ntotal = int(pos.shape[0])
        self.first = False
        self.stats["steps"] += 1
        self.stats["ghosts"] += ng

    # ------------------------------------------------------------------------------------------
    MIG_FIELDS = ("pos", "vel", "acc", "m", "T", "mu", "gam", "ptype", "E", "h", "gid")

    def _reorder_and_migrate(self):
        """Owned arrays into cell order, strays to their region's owner, arrivals appended: the kept
        particles are gathered once (one launch with a fused backend)."""
        s = self.s
        n = self.n_owned
        if not self.last or self.last["rho"].shape[0] != n:
            z = torch.zeros(n, dtype=torch.float64, device=self.device)
            self.last = dict(rho=z, nden=z.clone(), visc_heat=z.clone())
        mig = list(self.MIG_FIELDS) + list(self.extra_fields)
        fields = [s[k_] for k_ in mig] + [self.last[k_] for k_ in ("rho", "nden", "visc_heat")]
        order = self._cell_order()
        got = None
        sel = order
        if self.world > 1:
            owner = region_of(s["pos"], self.lo, self.hi)
            stay = owner == self.rank
            leave = torch.nonzero(~stay).flatten()             # usually a handful: group them by new owner
            own_l = owner[leave]
            o2 = torch.argsort(own_l, stable=True)
            leave = leave[o2]
            # counts per new owner, exchanged on the device; one host read serves both directions
            cnt_d = torch.bincount(own_l, minlength=self.world).to(self.comm_device)
            allc = [torch.zeros_like(cnt_d) for _ in range(self.world)]
            dist.all_gather(allc, cnt_d)
            both = torch.stack([cnt_d, torch.stack(allc)[:, self.rank]]).tolist()
            cnt, recv_counts = both[0], both[1]
            send_idx, o = [], 0
            for p in range(self.world):
                send_idx.append(None if p == self.rank else leave[o:o + cnt[p]])
                o += cnt[p]
            if leave.numel() or sum(recv_counts):
                got = self._exchange(send_idx, recv_counts, fields)
                self.stats["migrated"] += int(got.shape[0])
                if leave.numel():
                    sel = torch.nonzero(stay).flatten() if order is None else order[stay[order]]
        if sel is None and (got is None or got.shape[0] == 0):
            return                                             # nothing moved, no order to apply
        n_sel = n if sel is None else int(sel.numel())
        new = self._regroup(sel, n_sel, got, fields)
        for k_, v in zip(mig, new):
            s[k_] = v
        self.last = dict(rho=new[-3], nden=new[-2], visc_heat=new[-1])

    def owned_numpy(self):
        """This rank's owned particles as NumPy arrays keyed like the single-GPU download."""
        s = self.s
        c = lambda t: t.detach().cpu().numpy()
        out = dict(gid=c(s["gid"]), points=c(s["pos"]), velocities=c(s["vel"]), total_accel=c(s["acc"]),
                   E_internal=c(s["E"]), T=c(s["T"]), sizes=c(s["h"]), dt=self.dt_last)
        if self.last:
            out.update(densities=c(self.last["rho"]), num_densities=c(self.last["nden"]),
                       visc_heat=c(self.last["visc_heat"]))
        sp = getattr(self, "last_species", None)
        if sp:
            out.update({k_: c(v) for k_, v in sp.items() if v is not None})
        return out


    # ---- snapshot / restart and per-step diagnostics (for the decomposed run) -------------------------
    def snapshot(self, prefix):
        """Every rank writes `<prefix>.rank<r>.npz`: its owned particles (dynamic and static arrays, global ids) and what
        a restart needs of the driver's state (region boxes, step mode, first-step flag, last dt and mean / max h)."""
        s = self.s
        c = lambda t: t.detach().cpu().numpy()
        out = dict(gid=c(s["gid"]), points=c(s["pos"]), velocities=c(s["vel"]), total_accel=c(s["acc"]), E_internal=c(s["E"]),
                   T=c(s["T"]), sizes=c(s["h"]), mass=c(s["m"]), mu_array=c(s["mu"]), gamma_array=c(s["gam"]),
                   particle_type=c(s["ptype"]), lo=c(self.lo), hi=c(self.hi), first=np.int64(1 if self.first else 0),
                   dt_last=np.float64(self.dt_last), hmean_prev=np.float64(self.hmean_prev), hmax_prev=np.float64(self.hmax_prev),
                   world=np.int64(self.world), forms=np.array(self.forms), d=np.float64(self.d if self.d is not None else 0.0),
                   steps=np.int64(self.stats["steps"]),
                   # what the step does beyond the core sums, and the per-particle fields that go with it (they migrate
                   # with the particles: the snapshot holds them in the owned order like everything else)
                   with_drag=np.int64(self.with_drag), with_species=np.int64(self.with_species),
                   has_agb=np.int64(self.agb is not None))
        if self.with_drag:
            out.update(mean_grain_mass=c(s["mgm"]), mean_cross_section=c(s["mcs"]))
        if self.with_species:
            out["f_un"] = c(s["fun"])
        np.savez("%s.rank%d.npz" % (prefix, self.rank), **out)
        if self.world > 1:
            dist.barrier()

    @classmethod
    def from_snapshot(cls, prefix, backend, rank=0, world=1, **kw):
        """Resume a decomposed run written by snapshot() with the same number of ranks.  The step's extras come back as
        the run had them (with_drag / with_species default to the file's flags; the composition rows and the drag
        coefficients are in the file); an AGB table is not stored - a run that had one must be handed it again (agb=...)."""
        z = dict(np.load("%s.rank%d.npz" % (prefix, rank), allow_pickle=False))
        if int(z["world"]) != world:
            raise ValueError("snapshot was written by %d ranks, resuming with %d" % (int(z["world"]), world))
        state = {k_: z[k_] for k_ in ("points", "velocities", "total_accel", "E_internal", "T", "mass", "mu_array",
                                       "gamma_array", "particle_type", "gid")}
        for k_ in ("f_un", "mean_grain_mass", "mean_cross_section"):
            if k_ in z:
                state[k_] = z[k_]
        kw.setdefault("with_drag", bool(int(z.get("with_drag", 0))))
        kw.setdefault("with_species", bool(int(z.get("with_species", 0))))
        if int(z.get("has_agb", 0)) and kw.get("agb") is None:
            raise ValueError("the snapshot's run carried an AGB table (metallicity + yields in the species pass): pass agb=(splines, "
                             "mapto, divisor) to from_snapshot")
        forms = str(z["forms"])
        sim = cls(state, z["lo"], z["hi"], backend, rank, world, forms=forms, d=(float(z["d"]) if forms == "loop" else None), **kw)
        sim.s["h"] = torch.as_tensor(np.ascontiguousarray(z["sizes"], dtype=np.float64)).to(sim.device)
        sim.first = bool(int(z["first"]))
        sim.dt_last = float(z["dt_last"])
        sim.hmean_prev, sim.hmax_prev = float(z["hmean_prev"]), float(z["hmax_prev"])
        sim.stats["steps"] = int(z["steps"])
        return sim

    FAILURE_COUNTERS = ("bad_accel", "bad_energy", "bad_state", "bad_h", "short_rows")

    def failures(self, log_new=True):
        """Global failure counters (include/sphx.h sphx_stats.bad_*: owned particles whose acceleration / energy / updated
        state / kNN radius was NaN, inf or 0 before the reference's nan_to_num guards; searches that gave up short), summed
        over the ranks since the backend's stats were reset: one all_reduce.  Reading them waits for the stream, so the
        step does not do it by itself - call this where a driver prints its diagnostics (drv:465-468).  New ones are
        logged (Python logging, 'sph_code_amd')."""
        ctx = getattr(self.backend, "ctx", None)
        st = ctx.stats() if ctx is not None else {}
        loc = torch.tensor([float(st.get(k_, 0)) for k_ in self.FAILURE_COUNTERS], dtype=torch.float64, device=self.comm_device)
        if self.world > 1:
            dist.all_reduce(loc, op=dist.ReduceOp.SUM)
        out = {k_: int(v) for k_, v in zip(self.FAILURE_COUNTERS, loc.tolist())}
        seen = getattr(self, "_failures_seen", None) or {}
        new = {k_: v - seen.get(k_, 0) for k_, v in out.items() if v > seen.get(k_, 0)}
        self._failures_seen = out
        if new and log_new and self.rank == 0:
            import logging
            logging.getLogger("sph_code_amd").warning("decomposed step: non-finite values met (particles, all ranks): %s  "
                                                      "[cumulative: %s]", new, out)
        return out

    def diagnostics(self):
        """Global mass-weighted net acceleration (what sph/code_running.py:465-468 prints every step), momentum, kinetic
        and internal energy, particle count: one all_reduce(sum) of 9 numbers."""
        s = self.s
        m = s["m"]
        loc = torch.cat([(s["acc"] * m[:, None]).sum(dim=0), (s["vel"] * m[:, None]).sum(dim=0),
                         torch.stack([m.sum(), 0.5 * (m * (s["vel"] ** 2).sum(dim=1)).sum(), s["E"].sum(),
                                      torch.tensor(float(self.n_owned), dtype=torch.float64, device=m.device)])])
        if self.world > 1:
            loc = loc.to(self.comm_device)
            dist.all_reduce(loc, op=dist.ReduceOp.SUM)
        v = loc.tolist()
        return dict(net_accel=np.array(v[0:3]) / v[6], momentum=np.array(v[3:6]), kinetic=v[7], internal=v[8],
                    particles=int(round(v[9])), dt=self.dt_last)


def decompose_state(state, world, rank):
    """Global state dict -> (this rank's owned state incl. 'gid', lo, hi)."""
    owner, lo, hi = rcb_regions(state["points"], world)
    sel = np.nonzero(owner == rank)[0]
    mine = {}
    for key in ("points", "velocities", "mass", "particle_type", "T", "mu_array", "gamma_array", "E_internal",
                "total_accel"):
        mine[key] = np.ascontiguousarray(state[key][sel])
    if state.get("f_un") is not None:
        mine["f_un"] = np.ascontiguousarray(np.asarray(state["f_un"], dtype=np.float64)[sel])
    mine["gid"] = sel.astype(np.int64)
    return mine, lo, hi


# ==============================================================================================
# bench leg for N > 1 (called by bench.py under torch.distributed.run)
# ==============================================================================================
def _blob_split(be, sim):
    if not (getattr(sim, "overlap", False) and hasattr(be, "blob_split_counts")):
        return None
    try:
        c = be.blob_split_counts()
    except RuntimeError:
        return None
    return {"interior": c[0], "boundary": c[1], "idle": c[2], "overlap": "on" if sim.world > 1 else "one rank: nothing to overlap"}


def bench_main(args, rank, local_rank, world):
    from . import ics
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs GPUs (libsphx has no CPU path)")
    # SPHX_DIST_BACKEND=gloo rehearses the launch path with several ranks sharing one GPU (the
    # halo then travels through host memory); the default is RCCL over xGMI, one GPU per rank.
    backend = os.environ.get("SPHX_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world == 1:                               # (SPHX_FORCE_DIST=1 without a launcher: a rendezvous with itself)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # (a free one: a fixed port may still be held by an earlier run)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
    # an explicit timeout on every collective / p2p wait: a rank that hangs (a peer died, a link did not come up) fails
    # with the phase it was in instead of holding the node until the launcher's own limit
    import datetime
    tmo = datetime.timedelta(seconds=int(os.environ.get("SPHX_DIST_TIMEOUT_S", "300")))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        comm_dev = dev
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        comm_dev = torch.device("cpu")

    def give_up(where, exc):
        # never a re-exec, never a retry: say which rank failed where, and leave with a non-zero code (the launcher
        # then ends the other ranks)
        print("[sphx bench] rank %d/%d failed in %s (DistributedSim phase: %s): %s: %s"
              % (rank, world, where, DistributedSim.phase, type(exc).__name__, exc), file=sys.stderr, flush=True)
        os._exit(3)
    n_global = args.n * world                       # weak scaling: fixed particles per GPU
    forms = getattr(args, "forms", None) or "hydro_update"
    if forms == "loop" and args.workload in ("uniform_cube", "two_phase"):
        # BASELINE configs[3] as written: the reference's own IC (uniform cube of side 1.25e6 AU, drv:62,132) at its
        # natural size, stepped by the loop forms its time loop calls (drv:451-458) - stable, no dilution
        scale, dilution = 1.0, "none (natural size)"
    else:
        scale = ics.bench_size_scale(n_global)      # keeps the reference's hydro_update scheme in its stable regime
        dilution = "x%.3g in length (ics.bench_size_scale)" % scale
        if getattr(args, "natural_size", False) and n_global > 1e6:
            scale = (n_global / 1e6) ** (1. / 3.)
            dilution = "constant density: x%.3g in length" % scale
    species, drag = bool(getattr(args, "species", False)), bool(getattr(args, "drag", False))
    full = species or drag                           # these need the (N,15) composition
    # Box-filling workloads at their natural size (the default of N > 1: uniform_cube / two_phase): every rank draws ITS
    # OWN brick's particles (ics.cube_slab, seeded per brick) - no rank builds or bisects the global state (8 x 8e6
    # particles on an 8-GPU node otherwise).  The others: the global seeded IC, cut by recursive bisection, as before.
    slab = args.workload in ("uniform_cube", "two_phase") and scale == 1.0 and os.environ.get("SPHX_BENCH_GLOBAL_IC") != "1"
    state = None
    if not slab:
        state = (ics.WORKLOADS[args.workload](n_global, size_scale=scale) if full
                 else ics.WORKLOADS[args.workload](n_global, light=True, size_scale=scale))
        if full and state.get("f_un") is None:
            raise SystemExit("--species / --drag need a workload that carries f_un (two_phase, dusty_sphere)")
    agb_table = None
    if species:
        from . import agb as agb_mod
        z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "agb_reference.npz"))
        agb_table = agb_mod.splines_from_arrays(z["tx"], z["ty"], z["coeffs"], z["mapto"], float(z["divisor"]))
    mine = lo = hi = None
    if slab:
        side = 1.25e6 * ics.AU
        lo, hi, flo, fhi = brick_regions(world, [-0.5 * side] * 3, [0.5 * side] * 3)
        mine = ics.cube_slab(args.workload, args.n, rank, flo[rank], fhi[rank], light=not full)
        mine["gid"] = rank * args.n + np.arange(args.n, dtype=np.int64)
        mine = {k_: v for k_, v in mine.items() if v is not None}
    d_loop = None
    if forms == "loop":
        # one d for all ranks (drv:67-68 is a global): rank 0's estimate - on its own brick when the IC is drawn per rank
        # (the same density everywhere) - handed to the others
        dl = torch.zeros(1, dtype=torch.float64, device=comm_dev)
        if rank == 0:
            dl[0] = ics.loop_d(mine if slab else state, args.k)
        if world > 1:
            dist.broadcast(dl, src=0)
        d_loop = float(dl[0])
    # the same workload at the per-GPU size on ONE GPU through the fused single-GPU loop (rank 0, before the decomposed
    # run): the denominator a scaling efficiency for THIS workload needs, measured in the same process
    single = None
    if rank == 0 and world > 1 and os.environ.get("SPHX_BENCH_SINGLE", "1") != "0":
        from .sim import Simulation
        sc1 = 1.0 if dilution.startswith("none") else ics.bench_size_scale(args.n)
        s1 = ics.WORKLOADS[args.workload](args.n, size_scale=sc1) if full else ics.WORKLOADS[args.workload](args.n, light=True, size_scale=sc1)
        sim1 = Simulation(s1, n_neigh=args.k, device=dev_index, forms=forms, d=ics.loop_d(s1, args.k) if forms == "loop" else None,
                          clip_grad=getattr(args, "clip_grad", False), with_species=species, agb=agb_table, with_drag=drag)
        sim1.step(3)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sim1.step(10)
        torch.cuda.synchronize()
        el1 = time.perf_counter() - t1
        single = {"ms_per_step": el1 / 10 * 1e3, "particles": args.n, "particle_steps_per_s": args.n * 10 / el1}
        sim1.ctx.close()
        del sim1, s1
    if not slab:
        mine, lo, hi = decompose_state(state, world, rank)
    del state
    be = LibBackend(dev_index, k=args.k, clip_grad=getattr(args, "clip_grad", False))
    try:
        sim = DistributedSim(mine, lo, hi, be, rank, world, device=dev, comm_device=comm_dev, forms=forms, d=d_loop,
                             with_drag=drag, with_species=species, agb=agb_table)
        for _ in range(args.warmup):
            sim.step()
    except Exception as exc:                  # (RuntimeError of a timed-out / failed transfer, a library error ...)
        give_up("set-up / warm-up steps", exc)
    sim.ex.bytes_sent = 0
    sim.stats.update(ghosts=0, redo=0, migrated=0, replans=0)
    sim.host_ms.clear()
    be.ctx.reset_stats()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    try:
        for _ in range(args.steps):
            sim.step()
        torch.cuda.synchronize()
        dist.barrier()
    except Exception as exc:
        give_up("the timed steps", exc)
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([float(sim.n_owned), float(sim.stats["ghosts"]) / max(args.steps, 1),
                        float(sim.ex.bytes_sent) / max(args.steps, 1), float(sim.stats["redo"])],
                       dtype=torch.float64, device=comm_dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    v2 = (sim.s["vel"] ** 2).sum(dim=1)
    chk = torch.stack([torch.sqrt(v2.max()) if v2.numel() else v2.new_zeros(()),
                       (~torch.isfinite(sim.s["pos"])).sum().to(torch.float64)]).to(comm_dev)
    dist.all_reduce(chk, op=dist.ReduceOp.MAX)
    if os.environ.get("SPHX_MG_DEBUG") and rank == 0:
        q = torch.tensor([0.5, 0.9, 0.99, 1.0], dtype=torch.float64, device=sim.w_plan.device)
        print("[mg debug] coarse cell %.4g; h quantiles %s; claimed reach w quantiles %s; |v| dt quantiles %s; dt %.4g; owned %d ghosts/step %.0f"
              % (sim.g_cs, torch.quantile(sim.s["h"], q).tolist(), torch.quantile(sim.w_plan, q).tolist(),
                 (torch.quantile(torch.sqrt((sim.s["vel"] ** 2).sum(dim=1)), q) * sim.dt_last).tolist(), sim.dt_last,
                 sim.n_owned, sim.stats["ghosts"] / max(args.steps, 1)), file=sys.stderr, flush=True)
    fails = sim.failures(log_new=False)      # (a collective: every rank)
    kst = be.ctx.stats()                     # rank 0's kNN launches: HIP events on the stream they ran on
    if rank == 0:
        t = float(tmax[0])
        total = float(cnt[0])
        value = total * args.steps / t
        out = {
            "metric": "particle-steps/sec + achieved HBM GB/s, 10^6-particle 3D sphere, 1/2/4/8 GPUs",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": t / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, N=%d (%d per GPU), K=%d, fp64, poly6 kernel, viscosity on"
                                   % (args.workload, int(total), args.n, args.k),
                       "particles_per_gpu": args.n,
                       "decomposition": "recursive coordinate bisection, %d regions, %d-phase p2p halo" % (world, 3 if forms == "loop" else 4),
                       "backend": backend, "cloud_size_scale": scale, "dilution": dilution,
                       "forms": forms + (" (the reference's time loop: nsc.density, del_pressure, artificial_viscosity ...)" if forms == "loop" else ""),
                       "species_pass": species, "drag": drag,
                       "initial_condition": ("drawn per rank: each rank fills its own brick of the cube (ics.cube_slab)" if slab
                                             else "global seeded IC on every rank, cut by recursive coordinate bisection"),
                       "dist_timeout_s": int(tmo.total_seconds()),
                       "build": be.ctx.build_info(),
                       "env": {k_: v for k_, v in sorted(os.environ.items()) if k_.startswith("SPHX_")}},
            "single_gpu_same_workload": single,
            # value / (N x the SAME workload's single-GPU rate above): the efficiency of THIS workload's decomposition.
            # (The N = 1 line of `bench.py` is BASELINE configs[1] - the polytrope under hydro_update's sums - a different
            # workload: a curve through both is not a scaling curve.  `bench.py --gpus 1 --workload uniform_cube --forms
            # loop` prints the single-GPU line of this one.)
            "scaling_efficiency_same_workload": (value / (world * single["particle_steps_per_s"])) if single else None,
            "state_check": {"finite": float(chk[1]) == 0.0, "max_speed_m_s": float(chk[0]),
                            "max_drift_per_step_in_mean_h": float(chk[0]) * sim.dt_last / max(sim.hmean_prev, 1e-300),
                            # the library's failure counters over the timed steps, all ranks (sphx_stats.bad_*, by ballot in
                            # the update kernels: what the reference's nan_to_num guards hid)
                            "failures": fails},
            "roofline": (lambda ms, q: {"bound": "hbm", "kernel": "search (rank 0): knn_group_kernel + knn_kernel<0,2,1> (list mode)", "achieved": 192.0 * q / (ms * 1e-3) / 1e9,
                                        "peak": 8000.0, "unit": "GB/s", "frac": 192.0 * q / (ms * 1e-3) / 1e9 / 8000.0,
                                        "traffic": None,
                                        "traffic_source": "not collected for N > 1: the PMC passes run on the single-GPU bench (profiles/r02_final_pmc_per_launch.json)",
                                        "algorithmic_bytes_per_launch": 192.0 * q,
                                        "kernel_ms": ms, "queries_per_launch": q})(
                kst["ms_search"] / max(kst["search_steps"], 1), sim.n_owned) if kst["search_steps"] else None,
            "halo": {"ghosts_per_step_all_ranks": float(cnt[1]), "bytes_sent_per_step_all_ranks": float(cnt[2]),
                     "search_redos": float(cnt[3]),
                     "replans_per_step_rank0": sim.stats.get("replans", 0) / max(args.steps, 1),
                     "migrated_rank0": sim.stats.get("migrated", 0),
                     # workgroups of 128 particles by what they need from other ranks (last step, rank 0): the sums of the
                     # interior ones run under the scalar halo phases
                     "blobs_rank0": _blob_split(be, sim)},
            # rank 0's host wall time per step by section of DistributedSim.step (no added syncs: GPU
            # work is charged to the section whose host code next waits for it)
            "host_ms_per_step_rank0": {k_: v / max(args.steps, 1) for k_, v in sorted(sim.host_ms.items())},
            "step_model": {"algorithmic_bytes_per_particle_step": 1248,
                           "achieved_GBs": 1248 * value / 1e9,
                           "frac_of_hbm_peak": 1248 * value / 1e9 / (8000.0 * world)},
        }
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
