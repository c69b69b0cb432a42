#!/usr/bin/env python3
"""Instruction mix of k_fast_strip per stage, weighted by measured issue rates -> a mix-weighted issue floor.

  python tools/isa_mix.py [stage_counts.json]

Compiles csrc/orbx_kernels.hip to ISA (hipcc --save-temps, no GPU needed), takes the instructions between the
"; FSM name{" / "; FSM name}" markers the kernel leaves (FS_MARK), classifies every VALU instruction by the issue rate
measured for its opcode / operand form on MI355X (profiles/r02_valu_ops2.txt, profiles/r03_valu_ops3.txt: G wave-instr/s
over the whole chip; forms that were not measured take their class's mean), and -- with the per-launch stage counts of
tools/fast_mix.py (gpurun_out/fast_stage_counts.json) -- prints the time the launch's VALU instructions need at those
rates.  Writes profiles/fast_mix.json, which bench.py reports as FAST's `limiter`."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monoorbslam3_amd", "csrc")


def rates():
    r = {}
    for f in ("r02_valu_ops2.txt", "r03_valu_ops3.txt"):
        for line in open(os.path.join(ROOT, "profiles", f)):
            m = re.match(r"(\w+)\s+[\d.]+ ms\s+([\d.]+) G wave-instr/s", line)
            if m:
                r[m.group(1)] = float(m.group(2))
    return r


R = rates()
CHEAP = (R["xor"] + R["and"] + R["or"] + R["add"] + R["sub"] + R["lshr"] + R["mov"] + R["bitop3"]) / 8
SLOW = (R["alignbit"] + R["minu"] + R["bfe"] + R["mbcnt"] + R["cmp32"] + R["cndmask64"] + R["mad24"] + R["lshl"]) / 8
# opcode (without encoding suffix) -> measured row
ROW = {"v_xor_b32": "xor", "v_and_b32": "and", "v_or_b32": "or", "v_add_u32": "add", "v_sub_u32": "sub", "v_subrev_u32": "subrev",
       "v_lshrrev_b32": "lshr", "v_lshlrev_b32": "lshl", "v_and_or_b32": "andor", "v_or3_b32": "or3", "v_bfi_b32": "bfi",
       "v_xad_u32": "xad", "v_lshl_add_u32": "lshladd", "v_alignbit_b32": "alignbit", "v_alignbyte_b32": "alignbyte",
       "v_min_u32": "minu", "v_max_u32": "maxu", "v_max_i32": "maxi", "v_min_i32": "maxi", "v_min3_u32": "min3u", "v_max3_u32": "max3u",
       "v_perm_b32": "perm", "v_bfe_u32": "bfe", "v_bcnt_u32_b32": "bcnt", "v_mbcnt_lo_u32_b32": "mbcnt", "v_mbcnt_hi_u32_b32": "mbcnthi",
       "v_mov_b32": "mov", "v_sad_u8": "sad", "v_not_b32": "not", "v_bitop3_b32": "bitop3", "v_mad_u32_u24": "mad24",
       "v_mul_u32_u24": "mul24", "v_lshl_or_b32": "lshlor", "v_add3_u32": "add3", "v_ashrrev_i32": "ashr", "v_cndmask_b32": "cndmask64"}
CHEAP_OPS = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32", "v_not_b32",
             "v_bitop3_b32", "v_ashrrev_i32"}


def classify(line):
    """-> (kind, rate or None): kind in valu_cheap / valu_slow / lds / salu / vmem / other"""
    t = line.split()
    op = t[0]
    if op.startswith("ds_"):
        return "lds", None
    if op.startswith("s_"):
        return "salu", None
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem", None
    if not op.startswith("v_"):
        return "other", None
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    args = line[len(t[0]):]
    srcs = args.split(",")[1:]
    sgpr_src = any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?)\b", a) for a in srcs)
    if base.startswith("v_cmp"):
        return "valu_slow", R["cmp64"] if op.endswith("e64") else R["cmp32"]
    if base == "v_readfirstlane_b32" or base.startswith("v_readlane"):
        return "valu_slow", SLOW
    if base in CHEAP_OPS and not op.endswith(("sdwa", "dpp")):
        if sgpr_src:
            return "valu_slow", R["and_sgpr"]   # a scalar-register operand costs a cheap op its rate (measured on v_and_b32)
        return "valu_cheap", R.get(ROW.get(base, ""), CHEAP)
    return "valu_slow", R.get(ROW.get(base, ""), SLOW)


def kernel_isa(name="k_fast_strip"):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
               "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-c", os.path.join(CSRC, "orbx_kernels.hip"), "-o", os.path.join(d, "k.o"),
               "--save-temps"]
        subprocess.run(cmd, cwd=d, check=True, stderr=subprocess.DEVNULL)
        s = open(os.path.join(d, "orbx_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    m = re.search(r"^(_Z\d+%s\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel" % name, s, re.S | re.M)
    return [l.strip() for l in m.group(2).splitlines()]


def main():
    lines = kernel_isa()
    stages, cur = {}, None
    whole = {"valu_cheap": 0, "valu_slow": 0, "lds": 0, "salu": 0, "vmem": 0, "other": 0}
    labels, since_label, prefix = {}, [], {}

    def add(seg, l):
        kind, rate = classify(l)
        seg["n"][kind] += 1
        if rate:
            seg["t"] += 1.0 / rate              # ns of chip time per wave-instruction (rate in G wave-instr/s)

    def close(seg):
        for l in prefix.pop(seg["name"], []):   # a rotated loop: the body's tail sits between the loop header and the end marker
            add(seg, l)
        stages.setdefault(seg["name"], []).append(seg)

    for i, l in enumerate(lines):
        m = re.match(r"; FSM (\w+)_(begin|end)", l)
        if m:
            if m.group(2) == "begin":
                if cur is not None:
                    close(cur)
                cur = {"name": m.group(1), "n": dict.fromkeys(whole, 0), "t": 0.0, "at": i}
            elif cur is not None and cur["name"] == m.group(1):
                close(cur)
                cur = None
            else:
                prefix[m.group(1)] = list(since_label)
            continue
        ml = re.match(r"(\.LBB\d+_\d+):", l)
        if ml:
            labels[ml.group(1)] = i
            since_label = []
            continue
        if not l or l.startswith((";", ".", "_")):
            continue
        kind, rate = classify(l)
        whole[kind] += 1
        since_label.append(l)
        if cur is not None:
            add(cur, l)
            t = l.split()
            if t[0].startswith("s_cbranch") or t[0] == "s_branch":
                tgt = t[-1]
                if tgt in labels and labels[tgt] < cur["at"]:   # the loop's back edge: the stage's body ends here
                    close(cur)
                    cur = None
    counts = None
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fast_stage_counts.json")
    if os.path.exists(path):
        counts = json.load(open(path))
        if "valu_wave_instr_per_launch" not in counts:      # SQ_INSTS_VALU of the FAST launches of a step (tools/pmc_valu.py)
            try:
                counts["valu_wave_instr_per_launch"] = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))["wave_instr_per_step"]["fast"]
            except Exception:
                pass
    out = {"rates_G_wave_instr_s": {"cheap_class_mean": round(CHEAP, 1), "slow_class_mean": round(SLOW, 1)}, "stages": {}}
    print("static instructions of the whole kernel:", whole)
    print("%-9s %6s %6s %6s %5s %5s   ns of chip time per call   (copies in the ISA)" % ("stage", "cheap", "slow", "VALU", "LDS", "SALU"))
    for name, copies in stages.items():
        c = max(copies, key=lambda x: x["n"]["valu_cheap"] + x["n"]["valu_slow"])   # the loop-body copy (drain copies are equal or shorter)
        v = c["n"]["valu_cheap"] + c["n"]["valu_slow"]
        print("%-9s %6d %6d %6d %5d %5d   %8.4f                   %d" % (name, c["n"]["valu_cheap"], c["n"]["valu_slow"], v, c["n"]["lds"],
                                                                       c["n"]["salu"], c["t"], len(copies)))
        out["stages"][name] = {"valu_cheap": c["n"]["valu_cheap"], "valu_slow": c["n"]["valu_slow"], "lds": c["n"]["lds"],
                               "salu": c["n"]["salu"], "ns_per_call_at_measured_rates": round(c["t"], 5)}
    if counts:
        calls = {"tile": counts["tile_loads"], "compass": counts["compass_steps"], "arc": counts["arc_batches"],
                 "score": counts["score_batches"], "nms": counts["nms_batches"]}
        known_t = sum(out["stages"][k]["ns_per_call_at_measured_rates"] * calls[k] for k in calls if k in out["stages"])
        known_n = sum((out["stages"][k]["valu_cheap"] + out["stages"][k]["valu_slow"]) * calls[k] for k in calls if k in out["stages"])
        known_cheap = sum(out["stages"][k]["valu_cheap"] * calls[k] for k in calls if k in out["stages"])
        total_n = counts.get("valu_wave_instr_per_launch")      # PMC SQ_INSTS_VALU of the launch, when known
        rest_n = max(total_n - known_n, 0) if total_n else 0
        mean_t = known_t / known_n
        floor_ms = (known_t + rest_n * mean_t) * 1e-6
        out.update({"stage_calls_per_launch": calls, "frames": counts["frames"], "valu_in_marked_stages": int(known_n),
                    "valu_total_per_launch": total_n, "cheap_fraction_marked": round(known_cheap / known_n, 3),
                    "issue_floor_ms_per_launch": round(floor_ms, 4),
                    "note": "sum over VALU wave-instructions of 1 / (measured chip-wide rate of the opcode and operand form); "
                            "instructions outside the marked stages (pass set-up, score map, output) priced at the marked mean"})
        print("stage calls per launch:", calls)
        print("VALU in marked stages %.1f M (cheap class %.0f %%), launch total %s; mix-weighted issue floor %.3f ms per launch"
              % (known_n / 1e6, 100 * known_cheap / known_n, ("%.1f M" % (total_n / 1e6)) if total_n else "unknown", floor_ms))
        sys.path.insert(0, ROOT)
        from monoorbslam3_amd._lib import kernels_sha16
        out["kernels_sha16"] = kernels_sha16()       # bench.py drops the file from its line once the kernel sources change
        json.dump(out, open(os.path.join(ROOT, "profiles", "fast_mix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
