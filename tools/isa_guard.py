#!/usr/bin/env python3
"""Compiles the step kernel with -save-temps and reports on its two lag loops (the plain one:
896 FP64 multiply-adds per iteration, the e-term one: 1408): instruction count, scratch
accesses, and the tightest vmcnt wait inside the loop.  A vmcnt(0) there means the window
prefetch is drained every iteration.  Used by tests/test_isa_cpu.py.
usage: isa_guard.py [workdir]  -> prints one JSON object"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def analyse(workdir):
    src = os.path.join(ROOT, "mcmc_amd", "csrc", "mcmc_hip.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "mcmc_amd", "csrc"), "-DMCMC_T=8",
           "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form", "-save-temps", "-c", src, "-o", os.path.join(workdir, "x.o")]
    subprocess.run(cmd, cwd=workdir, check=True, capture_output=True)
    asm = open(os.path.join(workdir, "mcmc_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    out = {}
    # (mangled-name prefix, instruction prefixes counted as work, {work per iteration: loop name})
    kernels = [("_Z14mh_step_kernelILi8ELb1ELi0ELb0EE", ("v_fmac_f64", "v_fma_f64", "v_add_f64"),
                {896: "plain_loop", 1408: "eterm_loop"}),
               ("_Z14mh_step_kernelILi8ELb1ELi1ELb0EE", ("v_pk_fma_f32",),       # fast mode, mcmc_model_f32.hip.h
                {512: "fast_plain_loop", 768: "fast_eterm_loop"})]
    for sym, ops, names in kernels:
        m = re.search(r"^" + sym + r"[^\n]*:.*?\.end_amdhsa_kernel", asm, re.S | re.M)
        body = m.group(0).split("\n")
        blocks, cur, label = [], None, None
        for line in body:
            lm = re.match(r"^(\.LBB\d+_\d+):", line)
            if lm:
                cur, label = [], lm.group(1)
                blocks.append(cur)
            elif cur is not None:
                t = line.strip()
                if t and not t.startswith((";", ".")):
                    cur.append(t)
                    if t.startswith("s_cbranch") and t.split()[-1] == label:
                        cur = None                       # the loop ends at its backward branch
        for ins in blocks:
            fma = sum(1 for t in ins if t.startswith(ops))
            name = names.get(fma)
            if name is None or len(ins) < 500:
                continue
            waits = [int(x) for t in ins for x in re.findall(r"vmcnt\((\d+)\)", t)]
            # a kernel holds the plain loop twice (the runs before and after the chunks with e-terms): keep the
            # WORSE figure of each kind, so that no instance escapes the guard
            cur_stats = {"instructions": len(ins), "ops": fma, "scratch": sum("scratch" in t for t in ins),
                         "sgpr_spill_moves": sum(t.startswith(("v_readlane", "v_writelane")) for t in ins),
                         "min_vmcnt": min(waits) if waits else None, "instances": 1}
            if name in out:
                prev = out[name]
                cur_stats = {"instructions": max(prev["instructions"], cur_stats["instructions"]), "ops": fma,
                             "scratch": max(prev["scratch"], cur_stats["scratch"]),
                             "sgpr_spill_moves": max(prev["sgpr_spill_moves"], cur_stats["sgpr_spill_moves"]),
                             "min_vmcnt": min(x for x in (prev["min_vmcnt"], cur_stats["min_vmcnt"]) if x is not None)
                             if (prev["min_vmcnt"] is not None or cur_stats["min_vmcnt"] is not None) else None,
                             "instances": prev["instances"] + 1}
            out[name] = cur_stats
        if sym.endswith("Li0ELb0EE"):
            m0 = m
    # the wave mapping (mcmc_wave_kernel.hip.h): its plain lag loop takes two groups of 16 lags x 7 curves per trip
    for fuse in ("1", "0"):
        mw = re.search(r"^_Z19mh_step_wave_kernelILb" + fuse + r"EEv[^\n]*:.*?\.end_amdhsa_kernel", asm, re.S | re.M)
        lines = [t.strip() for t in mw.group(0).split("\n") if t.strip() and not t.strip().startswith((";", "."))]
        st = {"scratch": sum("scratch_" in t for t in lines), "mfma": sum(t.startswith("v_mfma") for t in lines),
              "dpp_fmac": sum(t.startswith("v_fmac_f64_dpp") for t in lines),
              "dpp_mov": sum(t.startswith("v_mov_b64_dpp") for t in lines)}
        cur, label = None, None
        for line in mw.group(0).split("\n"):
            lm = re.match(r"^(\.LBB\d+_\d+):", line)
            if lm:
                cur, label = [], lm.group(1)
                continue
            t = line.strip()
            if cur is None or not t or t.startswith((";", ".")):
                continue
            cur.append(t)
            if t.startswith("s_cbranch") and t.split()[-1] == label:          # a single-block loop ends here
                n = sum(1 for x in cur if x.startswith("v_fmac_f64_dpp"))
                if fuse == "1" and n == 224:
                    waits = [int(x) for y in cur for x in re.findall(r"lgkmcnt\((\d+)\)", y)]
                    st["plain_loop"] = {"instructions": len(cur), "ops": n, "min_lgkmcnt": min(waits) if waits else None}
                cur = None
        out["wave_kernel_fused" if fuse == "1" else "wave_kernel_strict"] = st
    m = m0
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(0))
    out["vgprs"] = int(vg.group(1)) if vg else None
    return out


if __name__ == "__main__":
    wd = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="isa_guard_")
    os.makedirs(wd, exist_ok=True)
    print(json.dumps(analyse(wd)))
