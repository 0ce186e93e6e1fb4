"""ISA lint for the hand-scheduled loaders of conv_igemm.hip.

The two-stage loaders keep loads in flight that the compiler does not know about.  That is only safe when the compiler can never
touch the destination registers, so the in-flight data lives in accumulation registers (a0..a63) that only inline asm names.
This check reads the generated assembly (`hipcc -S --cuda-device-only`) and fails when, in any kernel,
  * an inline-asm global load writes ordinary vector registers (the compiler may copy them before the data has landed), or
  * an instruction the compiler emitted (outside ASMSTART / ASMEND) names one of the accumulation registers the inline asm uses
    (registers above them are the compiler's: it spills vector registers there when a kernel is tight), or
  * the kernel's metadata reserves fewer accumulation registers than the inline asm uses.
`python tools/check_async_loads.py file.s` prints the violations and exits non-zero when there are any."""
import re
import sys


def check(path):
    kernel, in_asm, bad, foreign = None, False, [], []
    max_acc = {}
    agpr_meta = {}
    meta_name = None
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
            continue
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        m = re.match(r"\.amdhsa_kernel\s+(\S+)", s)
        if m:
            meta_name = m.group(1)
            continue
        m = re.match(r"\.amdhsa_accum_offset\s+(\d+)", s)
        if m and meta_name:
            agpr_meta.setdefault(meta_name, {})["accum_offset"] = int(m.group(1))
            continue
        m = re.match(r"\.amdhsa_next_free_vgpr\s+(\d+)", s)
        if m and meta_name:
            agpr_meta.setdefault(meta_name, {})["next_free_vgpr"] = int(m.group(1))
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        code = s.split(";")[0]
        accs = [int(x) for x in re.findall(r"\ba(\d+)\b", code)]
        for lo, hi in re.findall(r"\ba\[(\d+):(\d+)\]", code):
            accs += [int(lo), int(hi)]
        if in_asm:
            if accs:
                max_acc[kernel] = max(max_acc.get(kernel, -1), max(accs))
            if code.split()[0].startswith("global_load") and re.match(r"global_load\w*\s+v", code):
                bad.append(f"{kernel}: line {ln}: inline-asm load into vector registers: `{code.strip()}`")
        elif accs:
            foreign.append((kernel, ln, min(accs), code.strip()))
    for k, ln, lo, code in foreign:
        if lo <= max_acc.get(k, -1):
            bad.append(f"{k}: line {ln}: compiler-emitted instruction names a register of the hidden loaders (a0..a{max_acc[k]}): `{code}`")
    for k, top in max_acc.items():
        meta = agpr_meta.get(k)
        if not meta or "accum_offset" not in meta:
            bad.append(f"{k}: no accumulation-register metadata found")
            continue
        have = meta["next_free_vgpr"] - meta["accum_offset"]
        if have < top + 1:
            bad.append(f"{k}: inline asm uses a0..a{top} but the kernel reserves {have} accumulation registers")
    return bad, max_acc


def main(path):
    bad, max_acc = check(path)
    for b in bad:
        print(b)
    print(f"kernels with hidden-register loaders: {len(max_acc)}; violations: {len(bad)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
