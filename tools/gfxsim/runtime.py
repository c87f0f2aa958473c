"""gfxsim.runtime — device memory, code objects and kernel dispatch for the interpreter in gfxsim.cpu.

Memory is ONE flat arena (a numpy uint8 view); "device pointers" are real addresses inside it, so host code that was
compiled against HIP can run unchanged on top of the fake runtime (fakehip.cpp) and hand its pointers to kernels.

Dispatch: the workgroups of a launch run one after the other (in order of their linear id), the wavefronts of a workgroup
round-robin in slices, parked at s_barrier until all have arrived.  `resident` > 1 interleaves that many workgroups (for
kernels whose workgroups wait for each other).  Test infrastructure only.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import asm
from .cpu import Wave, SimError, M32, CODE_HI, SHARED_HI, PRIVATE_HI, AR

U8 = np.uint8


class Memory:
    """flat arena; `base` is the address of byte 0"""

    def __init__(self, size=None, base=None, buffer=None):
        if buffer is not None:
            self.a = buffer
            self.base = base
        else:
            self.a = np.zeros(size, dtype=U8)
            self.base = base if base is not None else 0x10000000
        self.size = self.a.size
        self.top = 256
        self.allocs = {}
        self.shadow = None          # memcheck: one byte per arena byte (0 never allocated, 1 allocated, 2 written); None = off
        self.findings = {}          # (kind, kernel, line) -> [count, first address]
        self.where = ("", 0)        # (kernel, source line) of the access being made — set by the wave's run loop while memcheck is on
        self.race_log = None        # racecheck: [(kind, arena offsets, bytes per lane, agent, epoch, line)] of the launch in flight
        self.agent = (0, 0)         # (workgroup number * 64 + wavefront, barrier epoch of its workgroup) — set by the wave's run loop

    def _finding(self, kind, addr):
        k = (kind,) + self.where
        f = self.findings.get(k)
        if f is None:
            self.findings[k] = [1, int(addr)]
        else:
            f[0] += 1

    def check_read(self, i, n):
        """i: numpy array of arena offsets (first byte of each lane's access), n bytes each"""
        sh = self.shadow[i[:, None] + np.arange(n)]
        if (sh != 2).any():
            bad = np.nonzero((sh != 2).any(axis=1))[0][0]
            col = np.nonzero(sh[bad] != 2)[0][0]
            self._finding("read of never-allocated memory" if sh[bad][col] == 0 else "read of memory nobody wrote", self.base + int(i[bad]) + int(col))

    def check_write(self, i, n):
        idx = (i[:, None] + np.arange(n)).ravel()
        sh = self.shadow[idx]
        if (sh == 0).any():
            self._finding("write to never-allocated memory", self.base + int(idx[np.nonzero(sh == 0)[0][0]]))
            idx = idx[sh != 0]
        self.shadow[idx] = 2

    def alloc(self, n, align=256):
        p = (self.top + align - 1) // align * align
        if p + n > self.size:
            raise SimError("arena exhausted")
        self.top = p + n
        self.allocs[self.base + p] = n
        return self.base + p

    def view(self, addr, n):
        i = addr - self.base
        if i < 0 or i + n > self.size:
            raise SimError("address 0x%x (+%d) is outside the arena" % (addr, n))
        return self.a[i:i + n]

    def read_dwords(self, addr, n):
        if self.shadow is not None:
            self.check_read(np.array([addr - self.base], dtype=np.int64), 4 * n)
        return [int(x) for x in self.view(addr, 4 * n).view("<u4")] if addr % 4 == 0 else \
            [int.from_bytes(self.view(addr + 4 * j, 4).tobytes(), "little") for j in range(n)]

    def write_dwords(self, addr, vals):
        for j, v in enumerate(vals):
            self.view(addr + 4 * j, 4)[:] = np.frombuffer(int(v & M32).to_bytes(4, "little"), dtype=U8)

    def gather(self, addr, n):
        i = addr - self.base
        if i.min() < 0 or i.max() + n > self.size:
            bad = addr[(i < 0) | (i + n > self.size)][0]
            raise SimError("global read at 0x%x (+%d) is outside the arena" % (int(bad), n))
        if self.shadow is not None:
            self.check_read(i, n)
        if self.race_log is not None:
            self.race_log.append(("r", i.copy(), n, self.agent[0], self.agent[1], self.where[1]))
        return self.a[i[:, None] + AR[n]]

    def scatter(self, addr, data):
        n = data.shape[1]
        i = addr - self.base
        if i.min() < 0 or i.max() + n > self.size:
            bad = addr[(i < 0) | (i + n > self.size)][0]
            raise SimError("global write at 0x%x (+%d) is outside the arena" % (int(bad), n))
        if self.shadow is not None:
            self.check_write(i, n)
        if self.race_log is not None:
            self.race_log.append(("w", i.copy(), n, self.agent[0], self.agent[1], self.where[1]))
        self.a[(i[:, None] + AR[n]).ravel()] = data.ravel()


POISON = bool(os.environ.get("GFXSIM_POISON"))      # LDS starts as garbage, as on the device (leftovers of earlier workgroups), instead of zeros
_poison_rng = np.random.default_rng(0x5EED)


class WorkGroup:
    def __init__(self, wid, lds_bytes):
        self.id = wid
        self.lds = _poison_rng.integers(0, 256, lds_bytes, dtype=U8) if POISON else np.zeros(lds_bytes, dtype=U8)
        self.lds_def = None         # memcheck: which LDS bytes this workgroup has written
        self.race = None            # racecheck (cpu.Race)
        self.waves = []


class Runtime:
    def __init__(self, mem=None, arena_bytes=1 << 28):
        self.mem = mem if mem is not None else Memory(arena_bytes)
        self.modules = []
        self.kernels = {}        # name -> (module, Kernel)
        self.symaddr = {}        # (module index, symbol) -> address
        self.clock = 1000
        self.stats = {}          # kernel name -> {class: count}
        self.launches = []
        self.slice = 400
        self.max_steps_per_wave = 200_000_000
        self.resident = 1
        self.trace = None
        self.verbose = False
        self.racecheck = False
        self.sched = None           # schedule fuzzing: a numpy Generator — random slice lengths, wavefront order and 8 workgroups in flight

    # ---- code objects ------------------------------------------------------------------------------------------
    def load(self, text, name="<asm>"):
        mod = asm.Module(text, name)
        mod.index = len(self.modules)
        self.modules.append(mod)
        for sec, buf in mod.data.items():
            if not buf:
                continue
            base = self.mem.alloc(len(buf), 256)
            self.mem.view(base, len(buf))[:] = np.frombuffer(bytes(buf), dtype=U8)
            self.defined(base, len(buf))
            for sym, (s, off) in mod.datasym.items():
                if s == sec:
                    self.symaddr[(mod.index, sym)] = base + off
            mod.data[sec] = base          # remember where the section lives
        for sec, off, sym, add, n in mod.datareloc:
            target = self.symaddr.get((mod.index, sym))
            if target is None and sym in mod.labels:
                target = self.code_addr(mod, mod.labels[sym])
            if target is None:
                raise SimError("data relocation against unknown symbol " + sym)
            self.mem.view(mod.data[sec] + off, n)[:] = np.frombuffer(int(target + add).to_bytes(n, "little"), dtype=U8)
        for kname, k in mod.kernels.items():
            if k.entry is not None:
                self.kernels[kname] = (mod, k)
        return mod

    def load_file(self, path):
        with open(path) as f:
            return self.load(f.read(), os.path.basename(path))

    def code_addr(self, mod, pc):
        return (CODE_HI << 32) | (mod.index << 40) | (pc * 8)

    def code_pc(self, mod, addr):
        if (addr >> 48) != (CODE_HI >> 16) or ((addr >> 40) & 0xFF) != mod.index:
            raise SimError("jump to 0x%x is not a code address of this module" % addr)
        return (addr & ((1 << 40) - 1)) // 8

    def reloc(self, mod, name, kind, addend, pc):
        key = (mod.index, name)
        S = self.symaddr.get(key)
        if S is None:
            if name in mod.labels:
                S = self.code_addr(mod, mod.labels[name])
            else:
                raise SimError("relocation against unknown symbol " + name)
        if kind.startswith("rel32"):
            P = self.code_addr(mod, pc) + 4
            v = (S + addend - P) & ((1 << 64) - 1)
        else:
            v = (S + addend) & ((1 << 64) - 1)
        return (v & M32) if kind.endswith("lo") else (v >> 32) & M32

    def absolute(self, mod, name):
        if name in mod.absolute:
            return mod.absolute[name] & M32
        raise SimError("symbol %s has no value the model knows" % name)

    def defined(self, addr, n):
        if self.mem.shadow is not None:
            self.mem.shadow[addr - self.mem.base:addr - self.mem.base + n] = 2

    def memcheck_report(self):
        """[(kind, kernel, source line, count, first address)] — findings of the memcheck mode, most frequent first"""
        return sorted(((k[0], k[1], k[2], v[0], v[1]) for k, v in self.mem.findings.items()), key=lambda t: -t[3])

    # ---- memory access from waves -----------------------------------------------------------------------------
    def read(self, w, addr, n, flat):
        if flat:
            hi = addr >> 32
            if (hi == SHARED_HI).any():
                if not (hi == SHARED_HI).all():
                    raise SimError("a flat access mixes LDS and global addresses")
                a = addr & M32
                if a.min() < 0 or a.max() + n > w.lds.size:
                    raise SimError("flat LDS read out of range")
                if w.wg.lds_def is not None and not w.wg.lds_def[a[:, None] + np.arange(n)].all():
                    self.mem._finding("LDS read of bytes this workgroup has not written", int(a[0]))
                return w.lds[a[:, None] + AR[n]]
            if (hi == PRIVATE_HI).any():
                raise SimError("flat access to the private aperture is not modelled")
        return self.mem.gather(addr, n)

    def write(self, w, addr, data, flat):
        if flat:
            hi = addr >> 32
            if (hi == SHARED_HI).any():
                if not (hi == SHARED_HI).all():
                    raise SimError("a flat access mixes LDS and global addresses")
                a = addr & M32
                n = data.shape[1]
                if a.min() < 0 or a.max() + n > w.lds.size:
                    raise SimError("flat LDS write out of range")
                if w.wg.lds_def is not None:
                    w.wg.lds_def[(a[:, None] + np.arange(n)).ravel()] = True
                w.lds[(a[:, None] + AR[n]).ravel()] = data.ravel()
                return
            if (hi == PRIVATE_HI).any():
                raise SimError("flat access to the private aperture is not modelled")
        self.mem.scatter(addr, data)

    def read_int(self, w, a, n, flat):
        if flat and (a >> 32) == SHARED_HI:
            a &= M32
            return int.from_bytes(w.lds[a:a + n].tobytes(), "little")
        if self.mem.shadow is not None:
            self.mem.check_read(np.array([a - self.mem.base], dtype=np.int64), n)
        return int.from_bytes(self.mem.view(a, n).tobytes(), "little")

    def write_int(self, w, a, v, n, flat):
        b = np.frombuffer(int(v).to_bytes(n, "little"), dtype=U8)
        if flat and (a >> 32) == SHARED_HI:
            a &= M32
            w.lds[a:a + n] = b
        else:
            if self.mem.shadow is not None:
                self.mem.check_write(np.array([a - self.mem.base], dtype=np.int64), n)
            if self.mem.race_log is not None:
                self.mem.race_log.append(("a", np.array([a - self.mem.base], dtype=np.int64), n, self.mem.agent[0], self.mem.agent[1], self.mem.where[1]))
            self.mem.view(a, n)[:] = b

    # ---- dispatch ----------------------------------------------------------------------------------------------
    def pack_args(self, kernel, values, grid, block, dyn_lds):
        """values: per explicit argument either bytes or an int"""
        size = max(kernel.desc.get("kernarg_size", 0), kernel.meta.get(".kernarg_segment_size", 0), 8)
        buf = bytearray(size + 64)
        explicit = [a for a in kernel.args if not a.get(".value_kind", "").startswith("hidden_")]
        if len(values) != len(explicit):
            raise SimError("%s takes %d arguments, got %d" % (kernel.name, len(explicit), len(values)))
        for a, v in zip(explicit, values):
            off, n = a[".offset"], a[".size"]
            if isinstance(v, (bytes, bytearray, memoryview)):
                bv = bytes(v)[:n]
                buf[off:off + len(bv)] = bv
            else:
                buf[off:off + n] = (int(v) & ((1 << (8 * n)) - 1)).to_bytes(n, "little")
        for a in kernel.args:
            kind = a.get(".value_kind", "")
            if not kind.startswith("hidden_"):
                continue
            off, n = a[".offset"], a[".size"]
            val = 0
            for ax, i in (("x", 0), ("y", 1), ("z", 2)):
                if kind == "hidden_block_count_" + ax:
                    val = grid[i]
                elif kind == "hidden_group_size_" + ax:
                    val = block[i]
                elif kind == "hidden_remainder_" + ax:
                    val = 0
            if kind == "hidden_grid_dims":
                val = 1 + (grid[1] * block[1] > 1 or grid[2] * block[2] > 1) + (grid[2] * block[2] > 1)
            elif kind == "hidden_dynamic_lds_size":
                val = dyn_lds
            elif kind == "hidden_shared_base":
                val = SHARED_HI
            elif kind == "hidden_private_base":
                val = PRIVATE_HI
            buf[off:off + n] = (int(val) & ((1 << (8 * n)) - 1)).to_bytes(n, "little")
        return bytes(buf[:size])

    def launch(self, name, grid, block, args=None, dyn_lds=0, kernarg=None):
        """grid = workgroups per axis, block = threads per axis.  args: list of ints / bytes (explicit arguments), or kernarg: raw bytes"""
        if name not in self.kernels:
            raise SimError("kernel %s is not loaded" % name)
        mod, k = self.kernels[name]
        grid = tuple(grid) + (1,) * (3 - len(grid))
        block = tuple(block) + (1,) * (3 - len(block))
        if kernarg is None:
            kernarg = self.pack_args(k, args or [], grid, block, dyn_lds)
        ka = self.mem.alloc(len(kernarg) + 64, 256)        # (the compiler rounds its scalar loads of the segment up: s_load_dwordx8 over 7 pointers)
        self.mem.view(ka, len(kernarg))[:] = np.frombuffer(kernarg, dtype=U8)
        self.defined(ka, len(kernarg) + 64)
        d = k.desc
        lds_bytes = d.get("group_segment_fixed_size", 0) + dyn_lds
        threads = block[0] * block[1] * block[2]
        nwaves = (threads + 63) // 64
        st = self.stats.setdefault(name, {})
        total_wg = grid[0] * grid[1] * grid[2]
        self.launches.append((name, grid, block))
        if self.racecheck:
            self.mem.race_log = []
            self._wg_counter = 0
        pending = [(x, y, z) for z in range(grid[2]) for y in range(grid[1]) for x in range(grid[0])]
        active = []
        pi = 0
        resident = self.resident if self.sched is None else max(self.resident, 8)
        while pi < len(pending) or active:
            while pi < len(pending) and len(active) < resident:
                active.append(self._make_wg(mod, k, pending[pi], ka, block, threads, nwaves, lds_bytes))
                pi += 1
            order = list(active)
            if self.sched is not None:
                self.sched.shuffle(order)
            for wg in order:
                if self._run_wg(wg, (self.slice * 4 if self.sched is None else int(self.sched.integers(50, 3000))) if resident > 1 else None):
                    active.remove(wg)
                    for w in wg.waves:
                        for c, n in w.count.items():
                            st[c] = st.get(c, 0) + n
        # the kernarg block stays allocated (bump allocator); small
        if self.racecheck:
            self._analyse_races(name)
            self.mem.race_log = None

    def _analyse_races(self, name):
        """global memory, one launch: a byte touched by two different wavefronts, at least one of them writing, not both atomically, and
        not on two sides of a barrier of their common workgroup.  (Accesses of different workgroups are never ordered inside a launch.)"""
        log = self.mem.race_log
        if not log or "rocprim" in name:           # (rocprim's decoupled look-back scan synchronises through flags in global memory by design)
            return
        nent = sum(e[1].size * e[2] for e in log)
        if nent > 60_000_000:
            self.mem.findings[("racecheck skipped (launch too large)", name, 0)] = [1, 0]
            return
        def cat(kinds):
            sel = [e for e in log if e[0] in kinds]
            if not sel:
                z = np.zeros(0, dtype=np.int64)
                return z, z, z, z
            idx = np.concatenate([(e[1][:, None] + np.arange(e[2])).ravel() for e in sel])
            ag = np.concatenate([np.full(e[1].size * e[2], e[3], dtype=np.int64) for e in sel])
            ep = np.concatenate([np.full(e[1].size * e[2], e[4], dtype=np.int64) for e in sel])
            ln = np.concatenate([np.full(e[1].size * e[2], e[5], dtype=np.int64) for e in sel])
            return idx, ag, ep, ln
        def groups(idx, ag, ep, ln):
            """per distinct byte: (byte, min agent, max agent, min epoch, max epoch, a line)"""
            if not idx.size:
                z = np.zeros(0, dtype=np.int64)
                return z, z, z, z, z, z
            o = np.argsort(idx, kind="stable")
            idx, ag, ep, ln = idx[o], ag[o], ep[o], ln[o]
            first = np.nonzero(np.r_[True, idx[1:] != idx[:-1]])[0]
            return (idx[first], np.minimum.reduceat(ag, first), np.maximum.reduceat(ag, first), np.minimum.reduceat(ep, first),
                    np.maximum.reduceat(ep, first), ln[first])
        def unordered(amin, amax, emin, emax):
            """two different wavefronts, and no barrier of a common workgroup provably between them"""
            return (amin != amax) & (((amin >> 6) != (amax >> 6)) | (emin == emax))
        wi, wa, we, wl = cat("w")
        b, amin, amax, emin, emax, ln = groups(wi, wa, we, wl)
        bad = unordered(amin, amax, emin, emax)
        if bad.any():
            k = np.nonzero(bad)[0]
            self.mem.findings[("global race: two wavefronts write the same byte in one launch", name, int(ln[k[0]]))] = [int(k.size), int(self.mem.base + b[k[0]])]
        # plain write vs atomic update of the same byte by different wavefronts
        ai, aa, ae, al = cat("a")
        if ai.size and wi.size:
            ab, aamin, aamax, aemin, aemax, aln = groups(ai, aa, ae, al)
            common, ia, ib = np.intersect1d(b, ab, assume_unique=True, return_indices=True)
            if common.size:
                bad = unordered(np.minimum(amin[ia], aamin[ib]), np.maximum(amax[ia], aamax[ib]), np.minimum(emin[ia], aemin[ib]), np.maximum(emax[ia], aemax[ib]))
                if bad.any():
                    k = np.nonzero(bad)[0]
                    self.mem.findings[("global race: a byte is written plainly by one wavefront and atomically by another in one launch", name, int(aln[ib][k[0]]))] = \
                        [int(k.size), int(self.mem.base + common[k[0]])]
        # reads against writes (plain or atomic) of other wavefronts
        ri, ra, re_, rl = cat("r")
        if ri.size and (wi.size or ai.size):
            xb, xamin, xamax, xemin, xemax, xl = groups(np.concatenate([wi, ai]), np.concatenate([wa, aa]), np.concatenate([we, ae]), np.concatenate([wl, al]))
            rb, ramin, ramax, remin, remax, rln = groups(ri, ra, re_, rl)
            common, ia, ib = np.intersect1d(xb, rb, assume_unique=True, return_indices=True)
            if common.size:
                amin = np.minimum(xamin[ia], ramin[ib]); amax = np.maximum(xamax[ia], ramax[ib])
                emin = np.minimum(xemin[ia], remin[ib]); emax = np.maximum(xemax[ia], remax[ib])
                bad = unordered(amin, amax, emin, emax)
                if bad.any():
                    k = np.nonzero(bad)[0]
                    self.mem.findings[("global race: a byte is read by one wavefront and written by another in one launch", name, int(xl[ia][k[0]]))] = [int(k.size), int(self.mem.base + common[k[0]])]

    def _make_wg(self, mod, k, wid, ka, block, threads, nwaves, lds_bytes):
        d = k.desc
        wg = WorkGroup(wid, lds_bytes)
        if self.racecheck:
            wg.number = self._wg_counter
            self._wg_counter += 1
            wg.epoch = 1
        if self.mem.shadow is not None:
            wg.lds_def = np.zeros(lds_bytes, dtype=bool)
        if self.racecheck and nwaves > 1 and lds_bytes:
            from .cpu import Race
            wg.race = Race(lds_bytes)
        for wi in range(nwaves):
            w = Wave(self, mod, k, wg, wi)
            w.trace = self.trace
            s = 0
            if d.get("user_sgpr_private_segment_buffer", 0):
                s += 4
            if d.get("user_sgpr_dispatch_ptr", 0):
                w.S[s], w.S[s + 1] = 0, 0           # not modelled: reading through it would fault in the arena check
                s += 2
            if d.get("user_sgpr_queue_ptr", 0):
                s += 2
            if d.get("user_sgpr_kernarg_segment_ptr", 0):
                w.S[s], w.S[s + 1] = ka & M32, ka >> 32
                s += 2
            if d.get("user_sgpr_dispatch_id", 0):
                s += 2
            if d.get("user_sgpr_flat_scratch_init", 0):
                s += 2
            if d.get("user_sgpr_private_segment_size", 0):
                s += 1
            npre = d.get("user_sgpr_kernarg_preload_length", 0)
            if npre:
                off = d.get("user_sgpr_kernarg_preload_offset", 0) * 4
                vals = self.mem.read_dwords(ka + off, npre)
                for j in range(npre):
                    w.S[s + j] = vals[j]
                s += npre
            s = max(s, d.get("user_sgpr_count", s))
            for ax, key in enumerate(("system_sgpr_workgroup_id_x", "system_sgpr_workgroup_id_y", "system_sgpr_workgroup_id_z")):
                if d.get(key, 0):
                    w.S[s] = wid[ax]
                    s += 1
            if d.get("system_sgpr_workgroup_info", 0):
                s += 1
            tid = wi * 64 + np.arange(64)
            live = tid < threads
            tx = tid % block[0]
            ty = (tid // block[0]) % block[1]
            tz = tid // (block[0] * block[1])
            w.V[0] = np.where(live, tx | (ty << 10) | (tz << 20), 0).astype(np.uint32)
            if not live.all():
                w.set_exec(int(sum(1 << int(l) for l in np.nonzero(live)[0])))
            w.priv_bytes = d.get("private_segment_fixed_size", 0)
            w.scratch_mem = (lambda w=w: _scratch(w))
            wg.waves.append(w)
        return wg

    def _run_wg(self, wg, budget=None):
        """returns True when the workgroup has finished"""
        spent = 0
        while True:
            progressed = False
            alive = 0
            waves = wg.waves
            if self.sched is not None:
                waves = list(waves)
                self.sched.shuffle(waves)
            for w in waves:
                if w.done:
                    continue
                alive += 1
                if w.at_barrier:
                    continue
                n = w.run(self.slice if self.sched is None else int(self.sched.integers(1, 700)))
                spent += n
                progressed = progressed or n > 0
                if w.steps > self.max_steps_per_wave:
                    raise SimError("wave %d of workgroup %s exceeded %d instructions (deadlock?)" % (w.index, wg.id, self.max_steps_per_wave))
            if alive == 0:
                return True
            live = [w for w in wg.waves if not w.done]
            if live and all(w.at_barrier for w in live):
                for w in live:
                    w.at_barrier = False
                progressed = True
                if wg.race is not None:
                    wg.race.epoch += 1
                if self.racecheck:
                    wg.epoch += 1
            if not progressed:
                raise SimError("workgroup %s makes no progress" % (wg.id,))
            if budget is not None and spent >= budget:
                return False


def _scratch(w):
    if w.scratch is None:
        w.scratch = np.zeros((64, max(w.priv_bytes, 16) + 256), dtype=U8)
    return w.scratch


# ------------------------------------------------------------------------------------------------------------------
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm(src, out, flags=("-O3", "-std=c++17", "-DSZL_LAB=0"), arch="gfx950"):
    """hipcc --cuda-device-only -S: the device assembly of one translation unit (cached by mtime)"""
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([HIPCC, "--offload-arch=" + arch, "--cuda-device-only", "-S", *flags, src, "-o", out], stderr=subprocess.DEVNULL)
    return out
