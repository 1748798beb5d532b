"""A small wave-level emulator of the gfx950 instructions the hand-written kernels use (rtk_lane_hot.S, rtk_packet_hot.S).

TEST INFRASTRUCTURE: it lets the CPU test suite run the assembly kernels -- their control flow, exec masks, LDS stack,
queue atomics, address arithmetic -- on small scenes without a GPU, with every memory access bounds-checked (on the GPU
an out-of-range access is a fault that can take the node down) and a cap on executed instructions (a wave that never
finishes is a hang there). It reads the disassembly of the object file (llvm-objdump), so it runs what the assembler
produced, not the source text.

Arithmetic: float adds, multiplies, conversions, compares, min / max, double-precision multiplies and adds are IEEE
(numpy), i.e. bit-exact. v_fma_f32 / v_pk_fma_f32 are evaluated in double precision (single rounding in all but
double-rounding corner cases: the kernels only use them in conservative box tests and inside the divide sequence).
v_div_scale / v_div_fmas / v_div_fixup: the sequence the kernels use is the compiler's IEEE divide; v_div_fixup returns
the correctly rounded quotient of its numerator and denominator operands, the steps before it are not modelled exactly.
v_rcp_f32 is the correctly rounded reciprocal (the hardware's is within 1 ulp).
Waves of a workgroup run one after the other (the kernels' waves only share atomics on queue heads).
"""
import re
import struct
import subprocess

import numpy as np

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
M64 = (1 << 64) - 1
LANES = np.arange(64, dtype=np.uint64)


class EmuError(RuntimeError):
    pass


def disassemble(path):
    """-> {kernel name: ([(addr, op, [operands], {modifiers})], {label: addr})}"""
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], check=True, capture_output=True, text=True).stdout
    kernels, cur, labels = {}, None, None
    pending = []
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            name = m.group(2)
            if cur is None or not name.startswith(("L_", ".L")):
                cur = name
                kernels[cur] = ([], {})
            pending.append(name)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            addr = int(m.group(3), 16)
            for n in pending:
                kernels[cur][1][n] = addr
            pending = []
            op, rest = m.group(1), m.group(2)
            mods = {}
            for k, v in re.findall(r"\b(op_sel_hi|op_sel|neg_lo|neg_hi):\[([0-9,]+)\]", rest):
                mods[k] = [int(x) for x in v.split(",")]
            mo = re.search(r"\boffset:(-?(?:0x)?[0-9a-fA-F]+)", rest)
            if mo:
                mods["offset"] = int(mo.group(1), 0)
            md = re.search(r"\b(quad_perm:\[[0-9,]+\]|row_shl:\d+|row_shr:\d+|row_ror:\d+|row_mirror|row_half_mirror|row_bcast:\d+|wave_\w+:\d+)", rest)
            if md:
                mods["dpp"] = md.group(1)
                for k in ("row_mask", "bank_mask"):
                    mk = re.search(k + r":(0x[0-9a-fA-F]+|\d+)", rest)
                    mods[k] = int(mk.group(1), 0) if mk else 0xf
                mods["bound_ctrl"] = "bound_ctrl" in rest
            rest = re.split(r"\s+(?=op_sel|neg_lo|neg_hi|offset:|nt\b|sc0\b|sc1\b|glc\b|quad_perm|row_|wave_|bank_mask|bound_ctrl)", rest)[0]
            ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
            kernels[cur][0].append((addr, op, ops, mods))
    return kernels


class Memory:
    """Flat address space made of named numpy byte buffers; every access is range-checked."""

    def __init__(self):
        self.bufs = []          # (base, bytes array, name)
        self.next = 0x10000000

    def add(self, name, array):
        b = np.ascontiguousarray(array).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.next += (len(b) + 0xfffff) & ~0xfffff
        self.next += 0x100000
        self.bufs.append((base, b, name))
        return base

    def get(self, base):
        for b0, b, _ in self.bufs:
            if b0 == base:
                return b
        raise KeyError(base)

    def find(self, addr, size, what):
        for b0, b, name in self.bufs:
            if b0 <= addr and addr + size <= b0 + len(b):
                return b, addr - b0
        raise EmuError("%s of %d bytes at %#x is outside every buffer" % (what, size, addr))

    def load(self, addr, size, what="load"):
        b, o = self.find(addr, size, what)
        return bytes(b[o:o + size])

    def store(self, addr, data, what="store"):
        b, o = self.find(addr, len(data), what)
        b[o:o + len(data)] = np.frombuffer(data, dtype=np.uint8)


def _f32(u):
    return np.asarray(u, dtype=np.uint32).view(np.float32)


def _u32(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def _mask_of(boolarr):
    return int(np.bitwise_or.reduce(np.where(boolarr, np.uint64(1) << LANES, np.uint64(0))))


def _lanes_of(mask):
    return ((np.uint64(mask) >> LANES) & np.uint64(1)).astype(bool)


class Wave:
    def __init__(self, code, labels, mem, lds, kernarg_addr, wg_id, wave_in_wg, max_instructions=2_000_000):
        self.code = code
        self.index = {a: i for i, (a, _, _, _) in enumerate(code)}
        self.labels = labels
        self.mem = mem
        self.lds = lds
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.s = np.zeros(128, dtype=np.uint32)
        self.vcc = 0
        self.exec = M64
        self.scc = 0
        self.m0 = 0
        self.s[0] = kernarg_addr & 0xffffffff
        self.s[1] = kernarg_addr >> 32
        self.s[2] = wg_id
        self.v[0] = np.arange(64, dtype=np.uint32) + 64 * wave_in_wg
        self.max_instructions = max_instructions
        self.executed = 0
        self.counts = {}
        self.watch = {"global_load_dword": 0, "s_load_dwordx8": 0, "s_load_dwordx16": 0}      # executions of whole opcodes (tests count a kernel's loads by them)

    # ---------------------------------------------------------------- operand access
    def sget(self, tok):
        if tok == "vcc":
            return self.vcc
        if tok == "exec":
            return self.exec
        if tok == "m0":
            return self.m0
        if tok in ("vcc_lo", "exec_lo"):
            return (self.vcc if tok == "vcc_lo" else self.exec) & 0xffffffff
        if tok in ("vcc_hi", "exec_hi"):
            return (self.vcc if tok == "vcc_hi" else self.exec) >> 32
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            a, b = int(m.group(1)), int(m.group(2))
            return sum(int(self.s[a + i]) << (32 * i) for i in range(b - a + 1))
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return int(self.s[int(m.group(1))])
        return self.const(tok)

    @staticmethod
    def const(tok, as_f64=False):
        try:
            return int(tok, 0) & M64
        except ValueError:
            pass
        try:
            f = float(tok)
        except ValueError:
            raise EmuError("operand not understood: %r" % tok)
        if as_f64:
            return struct.unpack("<Q", struct.pack("<d", f))[0]
        return struct.unpack("<I", struct.pack("<f", f))[0]

    def sset(self, tok, val):
        if tok == "vcc":
            self.vcc = val & M64
        elif tok == "exec":
            self.exec = val & M64
        elif tok == "m0":
            self.m0 = val & 0xffffffff
        elif tok == "exec_lo":
            self.exec = (self.exec & ~0xffffffff) | (val & 0xffffffff)
        elif tok == "exec_hi":
            self.exec = (self.exec & 0xffffffff) | ((val & 0xffffffff) << 32)
        elif tok == "vcc_lo":
            self.vcc = (self.vcc & ~0xffffffff) | (val & 0xffffffff)
        elif tok == "vcc_hi":
            self.vcc = (self.vcc & 0xffffffff) | ((val & 0xffffffff) << 32)
        else:
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
            if m:
                a, b = int(m.group(1)), int(m.group(2))
                for i in range(b - a + 1):
                    self.s[a + i] = (val >> (32 * i)) & 0xffffffff
            else:
                m = re.fullmatch(r"s(\d+)", tok)
                if not m:
                    raise EmuError("bad scalar destination %r" % tok)
                self.s[int(m.group(1))] = val & 0xffffffff

    def src(self, tok, float_const=True):
        """32-bit source as uint32[64] (no modifiers)"""
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return self.v[int(m.group(1))]
        if re.fullmatch(r"s(\d+)|vcc_lo|vcc_hi|exec_lo|exec_hi|m0", tok):
            return np.full(64, self.sget(tok) & 0xffffffff, dtype=np.uint32)
        return np.full(64, self.const(tok) & 0xffffffff, dtype=np.uint32)

    def srcf(self, tok):
        """float source with |x| and -x modifiers"""
        neg = tok.startswith("-")
        if neg:
            tok = tok[1:]
        ab = tok.startswith("|")
        if ab:
            tok = tok.strip("|")
        f = _f32(self.src(tok)).copy()
        if ab:
            f = np.abs(f)
        if neg:
            f = -f
        return f

    def src64(self, tok, is_float=False):
        neg = tok.startswith("-")
        if neg:
            tok = tok[1:]
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            a = int(m.group(1))
            val = self.v[a].astype(np.uint64) | (self.v[a + 1].astype(np.uint64) << np.uint64(32))
        elif tok.startswith("s[") or tok in ("vcc", "exec"):
            val = np.full(64, self.sget(tok), dtype=np.uint64)
        else:
            val = np.full(64, self.const(tok, as_f64=is_float), dtype=np.uint64)
        if neg:
            val = val ^ np.uint64(1 << 63)
        return val

    def vset(self, tok, val, mask=None):
        lanes = _lanes_of(self.exec if mask is None else mask)
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            r = int(m.group(1))
            self.v[r] = np.where(lanes, np.asarray(val, dtype=np.uint32), self.v[r])
            return
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if not m:
            raise EmuError("bad vector destination %r" % tok)
        a, b = int(m.group(1)), int(m.group(2))
        val = np.asarray(val)
        if val.dtype == np.uint64 and b - a == 1:
            parts = [(val & np.uint64(0xffffffff)).astype(np.uint32), (val >> np.uint64(32)).astype(np.uint32)]
        else:
            parts = [np.asarray(p, dtype=np.uint32) for p in val]
        for i, p in enumerate(parts):
            self.v[a + i] = np.where(lanes, p, self.v[a + i])

    def vsetf(self, tok, f):
        self.vset(tok, _u32(np.asarray(f, dtype=np.float32)))

    # ---------------------------------------------------------------- execution
    def run(self):
        pc = 0
        code = self.code
        with np.errstate(all="ignore"):
            while True:
                if pc >= len(code):
                    raise EmuError("ran off the end of the code")
                addr, op, ops, mods = code[pc]
                self.executed += 1
                if self.executed > self.max_instructions:
                    raise EmuError("instruction budget exceeded (a wave that does not finish) at %#x %s" % (addr, op))
                self.counts[op[:2]] = self.counts.get(op[:2], 0) + 1
                if op in self.watch:
                    self.watch[op] += 1
                if op == "s_endpgm":
                    return
                nxt = self.step(addr, op, ops, mods)
                pc = pc + 1 if nxt is None else self.index[nxt]

    def target(self, tok, addr):
        if tok in self.labels:
            return self.labels[tok]
        try:                       # a branch to a local numeric label is printed as its word offset
            imm = int(tok, 0) & 0xffff
        except ValueError:
            raise EmuError("unknown branch target %r" % tok)
        return addr + 4 + 4 * (imm - 0x10000 if imm & 0x8000 else imm)

    def dpp_source(self, ctrl, bound_ctrl):
        """-> (source lane of every lane, lanes whose source exists): the DPP controls of GFX9"""
        lane = np.arange(64)
        row = lane & ~15
        ok = np.ones(64, dtype=bool)
        m = re.fullmatch(r"quad_perm:\[(\d),(\d),(\d),(\d)\]", ctrl)
        if m:
            perm = np.array([int(x) for x in m.groups()])
            src = (lane & ~3) + perm[lane & 3]
        elif ctrl == "row_mirror":
            src = row + 15 - (lane & 15)
        elif ctrl == "row_half_mirror":
            src = (lane & ~7) + 7 - (lane & 7)
        elif ctrl.startswith(("row_shl:", "row_shr:", "row_ror:")):
            n = int(ctrl.split(":")[1])
            if ctrl.startswith("row_shl:"):
                i = (lane & 15) + n
                ok = i < 16
            elif ctrl.startswith("row_shr:"):
                i = (lane & 15) - n
                ok = i >= 0
            else:
                i = ((lane & 15) - n) % 16
            src = row + (i % 16)
        elif ctrl == "row_bcast:15":
            src = row - 1
            ok = lane >= 16
        elif ctrl == "row_bcast:31":
            src = np.full(64, 31)
            ok = lane >= 32
        else:
            raise EmuError("DPP control not modelled: %s" % ctrl)
        src = np.where(ok, src, lane)
        ok = ok & _lanes_of(self.exec)[src]            # a source lane that is switched off counts as missing
        return src, ok

    def step(self, addr, op, ops, mods):
        if op.endswith("_dpp"):
            src, ok = self.dpp_source(mods["dpp"], mods["bound_ctrl"])
            tok = ops[1]
            pre = ""
            while tok[:1] in "-|":
                pre += tok[0]; tok = tok[1:]
            tok = tok.rstrip("|")
            vals = self.src(tok)[src]
            lane = np.arange(64)
            enabled = (((mods["row_mask"] >> (lane >> 4)) & 1) == 1) & (((mods["bank_mask"] >> ((lane >> 2) & 3)) & 1) == 1)
            if mods["bound_ctrl"]:
                vals = np.where(ok, vals, 0)
            else:
                enabled = enabled & ok
            self.v[255] = vals
            ops = [ops[0], pre + "v255" + ("|" if "|" in pre else "")] + list(ops[2:])
            saved = self.exec
            self.exec = saved & _mask_of(enabled)
            try:
                return self.step(addr, op[:-4], ops, {})
            finally:
                self.exec = saved
        base = re.sub(r"_e(32|64)$", "", op)
        if base.startswith("s_"):
            return self.salu(addr, base, ops, mods)
        if base.startswith("v_cmp_"):
            return self.vcmp(base, op, ops)
        if base.startswith("v_"):
            return self.valu(base, ops, mods)
        if base.startswith("global_"):
            return self.vmem(base, ops, mods)
        if base.startswith("ds_"):
            return self.ds(base, ops, mods)
        raise EmuError("instruction not modelled: %s" % op)

    def salu(self, addr, op, ops, mods):
        g, st = self.sget, self.sset
        if op in ("s_nop", "s_waitcnt"):
            return None
        if op == "s_branch":
            return self.target(ops[0], addr)
        if op.startswith("s_cbranch_"):
            cond = {"scc0": self.scc == 0, "scc1": self.scc == 1, "vccz": self.vcc == 0, "vccnz": self.vcc != 0,
                    "execz": self.exec == 0, "execnz": self.exec != 0}[op[len("s_cbranch_"):]]
            return self.target(ops[0], addr) if cond else None
        if op.startswith("s_load_dwordx") or op == "s_load_dword":
            n = int(op[len("s_load_dwordx"):]) if op != "s_load_dword" else 1
            a = g(ops[1]) + (g(ops[2]) & 0xffffffff if not ops[2].startswith("s") else g(ops[2]))
            data = self.mem.load(a, 4 * n, "scalar load")
            m = re.fullmatch(r"s\[(\d+):(\d+)\]|s(\d+)", ops[0])
            first = int(m.group(1) if m.group(1) is not None else m.group(3))
            for i in range(n):
                self.s[first + i] = struct.unpack_from("<I", data, 4 * i)[0]
            return None
        if op == "s_mov_b32":
            st(ops[0], g(ops[1]) & 0xffffffff)
        elif op == "s_mov_b64":
            v = g(ops[1])
            if re.fullmatch(r"-?\d+", ops[1]):
                v = int(ops[1]) & M64
            st(ops[0], v)
        elif op == "s_not_b64":
            r = ~g(ops[1]) & M64
            st(ops[0], r); self.scc = int(r != 0)
        elif op in ("s_and_b64", "s_or_b64", "s_andn2_b64", "s_xor_b64", "s_and_b32", "s_or_b32", "s_andn2_b32", "s_xor_b32"):
            a, b = g(ops[1]), g(ops[2])
            w = M64 if op.endswith("b64") else 0xffffffff
            kind = op[2:op.rindex("_")]
            r = {"and": a & b, "or": a | b, "andn2": a & ~b, "xor": a ^ b}[kind] & w
            st(ops[0], r); self.scc = int(r != 0)
        elif op in ("s_add_u32", "s_sub_u32", "s_addc_u32"):
            a, b = g(ops[1]) & 0xffffffff, g(ops[2]) & 0xffffffff
            if op == "s_add_u32":
                r = a + b; self.scc = int(r > 0xffffffff)
            elif op == "s_addc_u32":
                r = a + b + self.scc; self.scc = int(r > 0xffffffff)
            else:
                r = a - b; self.scc = int(b > a)
            st(ops[0], r & 0xffffffff)
        elif op == "s_min_u32":
            a, b = g(ops[1]) & 0xffffffff, g(ops[2]) & 0xffffffff
            st(ops[0], min(a, b)); self.scc = int(a < b)
        elif op == "s_max_u32":
            a, b = g(ops[1]) & 0xffffffff, g(ops[2]) & 0xffffffff
            st(ops[0], max(a, b)); self.scc = int(a > b)
        elif op == "s_lshl1_add_u32":
            r = ((g(ops[1]) & 0xffffffff) << 1) + (g(ops[2]) & 0xffffffff)
            st(ops[0], r & 0xffffffff); self.scc = int(r > 0xffffffff)
        elif op == "s_max_i32":
            a, b = g(ops[1]) & 0xffffffff, g(ops[2]) & 0xffffffff
            a = a - (1 << 32) if a >> 31 else a
            b = b - (1 << 32) if b >> 31 else b
            st(ops[0], max(a, b) & 0xffffffff); self.scc = int(a > b)
        elif op in ("s_lshl_b32", "s_lshr_b32"):
            a, b = g(ops[1]) & 0xffffffff, g(ops[2]) & 31
            r = (a << b) & 0xffffffff if op == "s_lshl_b32" else a >> b
            st(ops[0], r); self.scc = int(r != 0)
        elif op == "s_mul_i32":
            st(ops[0], (g(ops[1]) * g(ops[2])) & 0xffffffff)
        elif op == "s_mul_hi_u32":
            st(ops[0], ((g(ops[1]) & 0xffffffff) * (g(ops[2]) & 0xffffffff)) >> 32)
        elif op == "s_lshl_b64":
            r = (g(ops[1]) << (g(ops[2]) & 63)) & M64
            st(ops[0], r); self.scc = int(r != 0)
        elif op == "s_lshl4_add_u32":
            r = ((g(ops[1]) & 0xffffffff) << 4) + (g(ops[2]) & 0xffffffff)
            st(ops[0], r & 0xffffffff); self.scc = int(r > 0xffffffff)
        elif op == "s_bfe_u32":
            a, b = g(ops[1]) & 0xffffffff, g(ops[2]) & 0xffffffff
            r = (a >> (b & 31)) & ((1 << ((b >> 16) & 0x7f)) - 1)
            st(ops[0], r); self.scc = int(r != 0)
        elif op == "s_bitcmp1_b32":
            self.scc = ((g(ops[0]) & 0xffffffff) >> (g(ops[1]) & 31)) & 1
        elif op == "s_bcnt1_i32_b32":
            r = bin(g(ops[1]) & 0xffffffff).count("1")
            st(ops[0], r); self.scc = int(r != 0)
        elif op == "s_cselect_b32":
            st(ops[0], (g(ops[1]) if self.scc else g(ops[2])) & 0xffffffff)
        elif op == "s_cmov_b32":
            if self.scc:
                st(ops[0], g(ops[1]) & 0xffffffff)
        elif op == "s_getpc_b64":
            st(ops[0], addr + 4)
        elif op == "s_setpc_b64":
            return g(ops[0])
        elif op == "s_bcnt1_i32_b64":
            r = bin(g(ops[1]) & M64).count("1")
            st(ops[0], r); self.scc = int(r != 0)
        elif op.startswith("s_cmp_"):
            kind, ty = op[len("s_cmp_"):].rsplit("_", 1)
            a, b = g(ops[0]), g(ops[1])
            if ty in ("u32", "i32"):
                a &= 0xffffffff; b &= 0xffffffff
                if ty == "i32":
                    a = a - (1 << 32) if a >> 31 else a
                    b = b - (1 << 32) if b >> 31 else b
            self.scc = int({"eq": a == b, "lg": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[kind])
        else:
            raise EmuError("scalar instruction not modelled: %s" % op)
        return None

    def vcmp(self, base, op, ops):
        kind, ty = base[len("v_cmp_"):].rsplit("_", 1)
        dst = ops[0]
        a_t, b_t = ops[1], ops[2]
        if ty == "f32":
            a, b = self.srcf(a_t), self.srcf(b_t)
            un = np.isnan(a) | np.isnan(b)
            r = {"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "lg": (a < b) | (a > b), "o": ~un, "u": un,
                 "ngt": ~(a > b), "nlt": ~(a < b), "nge": ~(a >= b), "nle": ~(a <= b), "neq": ~(a == b)}[kind]
        else:
            a, b = self.src(a_t), self.src(b_t)
            if ty == "i32":
                a, b = a.view(np.int32), b.view(np.int32)
            r = {"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "ne": a != b}[kind]
        self.sset(dst, _mask_of(r) & self.exec)
        return None

    def valu(self, op, ops, mods):
        s, f = self.src, self.srcf
        if op == "v_mov_b32":
            self.vset(ops[0], s(ops[1]))
        elif op in ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_max_f32", "v_min_f32"):
            a, b = f(ops[1]), f(ops[2])
            if op == "v_add_f32":
                r = a + b
            elif op == "v_sub_f32":
                r = a - b
            elif op == "v_mul_f32":
                r = a * b
            elif op == "v_max_f32":
                r = np.fmax(a, b)
            else:
                r = np.fmin(a, b)
            self.vsetf(ops[0], r)
        elif op in ("v_max3_f32", "v_min3_f32"):
            fn = np.fmax if op == "v_max3_f32" else np.fmin
            self.vsetf(ops[0], fn(fn(f(ops[1]), f(ops[2])), f(ops[3])))
        elif op in ("v_fma_f32", "v_div_fmas_f32"):
            r = f(ops[1]).astype(np.float64) * f(ops[2]).astype(np.float64) + f(ops[3]).astype(np.float64)
            self.vsetf(ops[0], r.astype(np.float32))
        elif op == "v_fmac_f32":
            r = f(ops[1]).astype(np.float64) * f(ops[2]).astype(np.float64) + f(ops[0]).astype(np.float64)
            self.vsetf(ops[0], r.astype(np.float32))
        elif op == "v_div_scale_f32":
            self.vsetf(ops[0], f(ops[2]))
            self.sset(ops[1], 0)
        elif op == "v_div_fixup_f32":
            # D = quotient of numerator (src2) and denominator (src1), correctly rounded
            self.vsetf(ops[0], (f(ops[3]).astype(np.float64) / f(ops[2]).astype(np.float64)).astype(np.float32))
        elif op == "v_rcp_f32":
            self.vsetf(ops[0], (1.0 / f(ops[1]).astype(np.float64)).astype(np.float32))
        elif op in ("v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32"):
            nsrc = 3 if op == "v_pk_fma_f32" else 2
            sel = (mods.get("op_sel", [0, 0, 0]) + [0, 0, 0])[:3]
            sel_hi = (mods.get("op_sel_hi", [1, 1, 1]) + [1, 1, 1])[:3]
            neg_lo = (mods.get("neg_lo", [0, 0, 0]) + [0, 0, 0])[:3]
            neg_hi = (mods.get("neg_hi", [0, 0, 0]) + [0, 0, 0])[:3]
            res = []
            for half, (se, ng) in enumerate(((sel, neg_lo), (sel_hi, neg_hi))):
                vals = []
                for i in range(nsrc):
                    v64 = self.src64(ops[1 + i])
                    word = ((v64 >> np.uint64(32 * se[i])) & np.uint64(0xffffffff)).astype(np.uint32)
                    x = _f32(word)
                    vals.append(-x if ng[i] else x)
                if op == "v_pk_fma_f32":
                    r = (vals[0].astype(np.float64) * vals[1].astype(np.float64) + vals[2].astype(np.float64)).astype(np.float32)
                elif op == "v_pk_add_f32":
                    r = (vals[0] + vals[1]).astype(np.float32)          # float32 arithmetic: one IEEE rounding, as the instruction's
                else:
                    r = (vals[0] * vals[1]).astype(np.float32)
                res.append(_u32(r))
            self.vset(ops[0], res)
        elif op in ("v_add_u32", "v_sub_u32", "v_subrev_u32"):
            a, b = s(ops[1]).astype(np.uint64), s(ops[2]).astype(np.uint64)
            r = a + b if op == "v_add_u32" else (a - b if op == "v_sub_u32" else b - a)
            self.vset(ops[0], (r & np.uint64(0xffffffff)).astype(np.uint32))
        elif op in ("v_and_b32", "v_or_b32", "v_xor_b32"):
            a, b = s(ops[1]), s(ops[2])
            self.vset(ops[0], a & b if op == "v_and_b32" else (a | b if op == "v_or_b32" else a ^ b))
        elif op == "v_lshlrev_b32":
            self.vset(ops[0], ((s(ops[2]).astype(np.uint64) << (s(ops[1]) & 31).astype(np.uint64)) & np.uint64(0xffffffff)).astype(np.uint32))
        elif op == "v_lshrrev_b32":
            self.vset(ops[0], s(ops[2]) >> (s(ops[1]) & 31))
        elif op == "v_lshl_add_u32":
            r = (s(ops[1]).astype(np.uint64) << (s(ops[2]) & 31).astype(np.uint64)) + s(ops[3]).astype(np.uint64)
            self.vset(ops[0], (r & np.uint64(0xffffffff)).astype(np.uint32))
        elif op == "v_mul_u32_u24":
            r = (s(ops[1]) & 0xffffff).astype(np.uint64) * (s(ops[2]) & 0xffffff).astype(np.uint64)
            self.vset(ops[0], (r & np.uint64(0xffffffff)).astype(np.uint32))
        elif op == "v_mul_lo_u32":
            r = s(ops[1]).astype(np.uint64) * s(ops[2]).astype(np.uint64)
            self.vset(ops[0], (r & np.uint64(0xffffffff)).astype(np.uint32))
        elif op == "v_cndmask_b32":
            mask = self.vcc if len(ops) < 4 else self.sget(ops[3])
            self.vset(ops[0], np.where(_lanes_of(mask), s(ops[2]), s(ops[1])))
        elif op.startswith("v_cvt_f32_ubyte"):
            k = int(op[-1])
            self.vsetf(ops[0], ((s(ops[1]) >> (8 * k)) & 255).astype(np.float32))
        elif op == "v_cvt_f64_f32":
            self.vset(ops[0], f(ops[1]).astype(np.float64).view(np.uint64))
        elif op == "v_cvt_f32_f64":
            self.vsetf(ops[0], self.src64(ops[1], True).view(np.float64).astype(np.float32))
        elif op == "v_fma_f64":
            # (the kernels only use it where a * b is exact in double precision -- products of converted floats --, so a * b + c is
            # the single-rounded result)
            a, b, c = (self.src64(ops[i], True).view(np.float64) for i in (1, 2, 3))
            self.vset(ops[0], np.asarray(a * b + c, dtype=np.float64).view(np.uint64))
        elif op in ("v_mul_f64", "v_add_f64", "v_min_f64", "v_max_f64"):
            a, b = self.src64(ops[1], True).view(np.float64), self.src64(ops[2], True).view(np.float64)
            if op == "v_mul_f64":
                r = a * b
            elif op == "v_add_f64":
                r = a + b
            elif op == "v_min_f64":
                r = np.where(a < b, a, b)      # (the kernels' keys are never NaN: a plain ordered select)
            else:
                r = np.where(a < b, b, a)
            self.vset(ops[0], np.asarray(r, dtype=np.float64).view(np.uint64))
        elif op == "v_mbcnt_lo_u32_b32" or op == "v_mbcnt_hi_u32_b32":
            mask = s(ops[1]).astype(np.uint64)
            lane = LANES if op == "v_mbcnt_lo_u32_b32" else np.where(LANES >= 32, LANES - np.uint64(32), np.uint64(0))
            limit = np.where(LANES >= 32, np.uint64(32), LANES) if op == "v_mbcnt_lo_u32_b32" else lane
            below = (np.uint64(1) << limit) - np.uint64(1)
            cnt = np.array([bin(int(m & b)).count("1") for m, b in zip(mask, below)], dtype=np.uint32)
            self.vset(ops[0], cnt + s(ops[2]))
        elif op == "v_permlane32_swap_b32":
            # lanes 32-63 of the first operand <-> lanes 0-31 of the second (gfx950)
            ra, rb = (int(re.fullmatch(r"v(\d+)", o).group(1)) for o in ops[:2])
            a, b = self.v[ra].copy(), self.v[rb].copy()
            if self.exec != M64:
                raise EmuError("v_permlane32_swap_b32 with lanes switched off is not modelled")
            self.v[ra][32:] = b[:32]
            self.v[rb][:32] = a[32:]
        elif op == "v_readlane_b32":
            self.sset(ops[0], int(s(ops[1])[self.sget(ops[2]) & 63]))
        elif op == "v_writelane_b32":
            r = int(re.fullmatch(r"v(\d+)", ops[0]).group(1))
            self.v[r][self.sget(ops[2]) & 63] = self.sget(ops[1]) & 0xffffffff
        elif op == "v_readfirstlane_b32":
            if self.exec == 0:
                lane = 0
            else:
                lane = (self.exec & -self.exec).bit_length() - 1
            self.sset(ops[0], int(s(ops[1])[lane]))
        else:
            raise EmuError("vector instruction not modelled: %s" % op)
        return None

    def vmem(self, op, ops, mods):
        off = mods.get("offset", 0)
        lanes = np.nonzero(_lanes_of(self.exec))[0]
        if op.startswith("global_load_dword"):
            n = 1 if op == "global_load_dword" else int(op[len("global_load_dwordx"):])
            dst, voff, sbase = ops[0], ops[1], ops[2]
            basea = self.sget(sbase)
            first = int(re.match(r"v\[?(\d+)", dst).group(1))
            vo = self.src(voff)
            for l in lanes:
                data = self.mem.load(basea + int(vo[l]) + off, 4 * n, op)
                for i in range(n):
                    self.v[first + i][l] = struct.unpack_from("<I", data, 4 * i)[0]
        elif op.startswith("global_store_"):
            voff, data_t, sbase = ops[0], ops[1], ops[2]
            basea = self.sget(sbase)
            vo = self.src(voff)
            first = int(re.match(r"v\[?(\d+)", data_t).group(1))
            if op == "global_store_byte":
                for l in lanes:
                    self.mem.store(basea + int(vo[l]) + off, bytes([int(self.v[first][l]) & 255]), op)
            else:
                n = 1 if op == "global_store_dword" else int(op[len("global_store_dwordx"):])
                for l in lanes:
                    self.mem.store(basea + int(vo[l]) + off, b"".join(struct.pack("<I", int(self.v[first + i][l])) for i in range(n)), op)
        elif op in ("global_atomic_add_x2", "global_atomic_add"):
            # returning form (sc0): dst, voffset, data, saddr; without a return value: voffset, data, saddr
            returns = len(ops) == 4
            dst, voff, data_t, sbase = ops if returns else [None] + list(ops)
            basea = self.sget(sbase)
            vo = self.src(voff)
            wide = op.endswith("x2")
            first = int(re.match(r"v\[?(\d+)", dst).group(1)) if returns else None
            dfirst = int(re.match(r"v\[?(\d+)", data_t).group(1))
            for l in lanes:
                a = basea + int(vo[l]) + off
                if wide:
                    old = struct.unpack("<Q", self.mem.load(a, 8, op))[0]
                    add = int(self.v[dfirst][l]) | (int(self.v[dfirst + 1][l]) << 32)
                    self.mem.store(a, struct.pack("<Q", (old + add) & M64), op)
                    if returns:
                        self.v[first][l] = old & 0xffffffff
                        self.v[first + 1][l] = old >> 32
                else:
                    old = struct.unpack("<I", self.mem.load(a, 4, op))[0]
                    self.mem.store(a, struct.pack("<I", (old + int(self.v[dfirst][l])) & 0xffffffff), op)
                    if returns:
                        self.v[first][l] = old
        else:
            raise EmuError("memory instruction not modelled: %s" % op)
        return None

    def ds(self, op, ops, mods):
        off = mods.get("offset", 0)
        lanes = np.nonzero(_lanes_of(self.exec))[0]
        if op in ("ds_write_b64", "ds_write_b32"):
            n = 2 if op.endswith("b64") else 1
            addr = self.src(ops[0])
            first = int(re.match(r"v\[?(\d+)", ops[1]).group(1))
            for l in lanes:
                a = int(addr[l]) + off
                if a < 0 or a + 4 * n > len(self.lds):
                    raise EmuError("LDS write at %#x outside the %d-byte allocation" % (a, len(self.lds)))
                for i in range(n):
                    self.lds[a + 4 * i:a + 4 * i + 4] = np.frombuffer(struct.pack("<I", int(self.v[first + i][l])), dtype=np.uint8)
        elif op in ("ds_read_b64", "ds_read_b32"):
            n = 2 if op.endswith("b64") else 1
            addr = self.src(ops[1])
            first = int(re.match(r"v\[?(\d+)", ops[0]).group(1))
            for l in lanes:
                a = int(addr[l]) + off
                if a < 0 or a + 4 * n > len(self.lds):
                    raise EmuError("LDS read at %#x outside the %d-byte allocation" % (a, len(self.lds)))
                for i in range(n):
                    self.v[first + i][l] = struct.unpack("<I", bytes(self.lds[a + 4 * i:a + 4 * i + 4]))[0]
        else:
            raise EmuError("LDS instruction not modelled: %s" % op)
        return None


def run_kernel(obj_path, kernel, mem, kernarg_bytes, workgroups, lds_bytes, waves_per_wg=4, max_instructions=2_000_000):
    """Run `kernel` of the object file for `workgroups` workgroups of `waves_per_wg` waves; -> per-wave instruction class counts."""
    code, labels = disassemble(obj_path)[kernel]
    base = code[0][0]
    ka = mem.add("kernarg", np.frombuffer(kernarg_bytes, dtype=np.uint8))
    stats = []
    for wg in range(workgroups):
        lds = np.zeros(lds_bytes, dtype=np.uint8)
        for w in range(waves_per_wg):
            wave = Wave(code, labels, mem, lds, ka, wg, w, max_instructions)
            wave.run()
            stats.append(dict(wave.counts, total=wave.executed, **wave.watch))
    return stats
