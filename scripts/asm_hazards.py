#!/usr/bin/env python3
"""Wait-state linter for the hand-written gfx950 kernels (rtk_packet_hot.S, rtk_lane_hot.S).

The assembler pads nothing: a hazard the compiler would cover with `s_nop` is silent wrong data in hand-written code.
This script disassembles a code object and checks, along the straight-line order of the instructions (branches are not
followed: a label is simply the next instruction, which is the conservative reading for fall-through paths), the
gfx940-family rules that apply to this code (LLVM GCNHazardRecognizer):

  R1  VALU writes an SGPR / VCC        -> VALU reads it (operand, carry-in, v_cndmask mask)     2 wait states
  R2  transcendental (v_rcp_f32 ...)   -> VALU reads the result                                 1
  R3  VALU writes an SGPR              -> VMEM reads it (saddr)                                 5
  R4  VALU writes VCC                  -> v_div_fmas                                            4
  R5  VALU writes an SGPR              -> v_readlane / v_writelane lane select                  4
  R6  global / buffer store of > 8 B   -> VALU overwrites the data registers                    2
  R7  VALU writes a VGPR               -> v_readfirstlane / v_readlane reads it                 1
  R8  VALU writes a VGPR               -> a DPP instruction reads it                            2
  R9  VALU writes a VGPR               -> v_permlane16/32_swap reads it (either operand)        2

An instruction is one wait state, `s_nop N` is N + 1. Usage: asm_hazards.py <code object or .o> [...]; exit status 1 on a finding.
"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_", "v_rcp_iflag")


def regs_of(tok):
    """'s[4:5]' -> {('s',4),('s',5)}; 'v3' -> {('v',3)}; 'vcc' -> vcc pair; modifiers stripped."""
    tok = tok.strip().lstrip("-").strip("|")
    m = re.fullmatch(r"([sv])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([sv])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {("s", "vcc")}
    if tok == "exec":
        return {("s", "exec")}
    if tok == "m0":
        return {("s", "m0")}
    return set()


class Ins:
    def __init__(self, addr, text):
        self.addr = addr
        self.text = text
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = parts[1] if len(parts) > 1 else ""
        ops = re.split(r"\s+(?=op_sel|op_sel_hi|neg_lo|neg_hi|offset:|nt\b|sc0\b|sc1\b|glc\b|row_|quad_|clamp|mul:|div:)", ops)[0]
        self.operands = [o.strip() for o in ops.split(",")] if ops.strip() else []
        o = self.op
        self.is_valu = o.startswith("v_")
        self.is_vmem = o.startswith(("global_", "buffer_", "flat_", "scratch_"))
        self.is_store = self.is_vmem and ("store" in o)
        self.is_trans = o.startswith(TRANS)
        self.is_nop = o == "s_nop"
        self.is_dpp = "_dpp" in o
        self.states = (int(self.operands[0], 0) + 1) if self.is_nop else 1
        self.writes = set()
        self.reads = set()
        if self.is_valu:
            n_dst = 1
            e32_vcc_write = o.startswith("v_cmp") and o.endswith("_e32") or o in ("v_add_co_u32_e32", "v_sub_co_u32_e32", "v_subrev_co_u32_e32",
                                                                                   "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subbrev_co_u32_e32")
            if o.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")) or (o.endswith("_e64") and ("_co_" in o)):
                n_dst = 2
            self.is_swap = o.startswith(("v_permlane16_swap", "v_permlane32_swap"))
            if self.is_swap:
                n_dst = 0          # both operands are read AND written
            if o.startswith("v_cmp") and o.endswith("_e32"):
                n_dst = 0 if (self.operands and self.operands[0] != "vcc") else 1
            for d in self.operands[:n_dst]:
                self.writes |= regs_of(d)
            if e32_vcc_write:
                self.writes |= {("s", "vcc")}
            for s in self.operands[n_dst:]:
                self.reads |= regs_of(s)
            if self.is_swap:
                self.writes |= self.reads
            if o in ("v_cndmask_b32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subbrev_co_u32_e32") or o.startswith("v_div_fmas"):
                self.reads |= {("s", "vcc")}
            if o.startswith(("v_fmac", "v_mac")) and self.operands:
                self.reads |= regs_of(self.operands[0])
        elif self.is_vmem:
            if self.is_store:
                for s in self.operands:
                    self.reads |= regs_of(s)
            else:
                self.writes |= regs_of(self.operands[0]) if self.operands else set()
                for s in self.operands[1:]:
                    self.reads |= regs_of(s)


def disassemble(path):
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], check=True, capture_output=True, text=True).stdout
    funcs = {}
    cur = None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            # local labels (L_...) are part of the kernel they sit in: one instruction stream per global symbol
            if cur is None or not m.group(1).startswith(("L_", ".L")):
                cur = m.group(1)
                funcs[cur] = []
            continue
        m = re.match(r"^\s+([a-z_0-9]+.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            funcs[cur].append(Ins(int(m.group(2), 16), m.group(1).strip()))
    return funcs


def lint(name, ins):
    findings = []

    def look_back(i, max_states):
        """yield (instruction, wait states between it and instruction i) for producers within max_states"""
        gap = 0
        j = i - 1
        while j >= 0 and gap < max_states:
            yield ins[j], gap
            gap += ins[j].states
            j -= 1

    for i, c in enumerate(ins):
        if c.is_nop:
            continue
        sr = {r for r in c.reads if r[0] == "s"}
        vr = {r for r in c.reads if r[0] == "v"}
        for p, gap in look_back(i, 5):
            if p.is_nop:
                continue
            p_sw = {r for r in p.writes if r[0] == "s"} if p.is_valu else set()
            p_vw = {r for r in p.writes if r[0] == "v"} if p.is_valu else set()
            if c.is_valu and p_sw & sr:
                need = 4 if (c.op.startswith("v_div_fmas") and ("s", "vcc") in p_sw) else 2
                if c.op.startswith(("v_readlane", "v_writelane")) and (regs_of(c.operands[-1]) & p_sw):
                    need = 4
                if gap < need:
                    findings.append((c, p, "R1/R4/R5 VALU-written SGPR read by VALU", need, gap))
            if c.is_vmem and p_sw & sr and gap < 5:
                findings.append((c, p, "R3 VALU-written SGPR read by VMEM", 5, gap))
            if c.is_valu and p.is_trans and (p_vw & vr) and gap < 1:
                findings.append((c, p, "R2 transcendental result read by VALU", 1, gap))
            if c.is_valu and p.is_store and gap < 2:
                data = set()
                for o in p.operands:
                    r = regs_of(o)
                    if len(r) > 2 and all(x[0] == "v" for x in r):
                        data |= r
                if data & {r for r in c.writes if r[0] == "v"}:
                    findings.append((c, p, "R6 store data overwritten by VALU", 2, gap))
            if c.op.startswith(("v_readfirstlane", "v_readlane")) and (p_vw & vr) and gap < 1:
                findings.append((c, p, "R7 VALU-written VGPR read by readlane", 1, gap))
            if getattr(c, "is_swap", False) and (p_vw & vr) and gap < 2:
                findings.append((c, p, "R9 VALU-written VGPR read by a permlane swap", 2, gap))
            if c.is_dpp and (p_vw & vr) and gap < 2:
                findings.append((c, p, "R8 VALU-written VGPR read by a DPP instruction", 2, gap))
    for c, p, what, need, gap in findings:
        print("%s: %s: needs %d wait states, has %d\n    producer %#x  %s\n    consumer %#x  %s" % (name, what, need, gap, p.addr, p.text, c.addr, c.text))
    return len(findings)


def main():
    bad = 0
    for path in sys.argv[1:]:
        for name, ins in disassemble(path).items():
            n = lint(name, ins)
            print("%s: %s: %d instructions, %d findings" % (path, name, len(ins), n))
            bad += n
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
