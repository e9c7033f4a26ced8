"""Fills the @@...@@ placeholders of DESIGN.template.md from the committed evidence (profiles/r03_*) -> DESIGN.md."""
import csv, json, os
P = "profiles"
d = json.load(open(f"{P}/r03_bench.json"))
r, cb = d["roofline"], d["cpu_baseline"]
w = {k: json.load(open(f"{P}/r03_bench_{k}.json")) for k in ("dsg_phase_audioldm2", "mpgd_sr4", "diffmusic_style_audioldm2")}
s3, s4 = json.load(open(f"{P}/r03_bench_strong_dsg_g32_n1.json")), json.load(open(f"{P}/r03_bench_strong_mpgd_g16_n1.json"))
st = d["stage_ms"]
stages = (f"U-Net(2B)+CFG {st['unet_cfg']:.1f}, VAE fwd {st['vae_fwd']:.1f} / bwd {st['vae_bwd']:.1f}, HiFi-GAN fwd {st['hifigan_fwd']:.1f} / bwd "
          f"{st['hifigan_bwd']:.1f}, operator+mel+loss fwd+bwd {st['operator_mel_loss_fwd_bwd']:.2f}, update {st['sched_update']:.3f}")
def pmc(c):
    rows = [x for x in csv.DictReader(open(f"{P}/r03_pmc_{c}_per_kernel.csv"))]
    tot = sum(float(x[f"{c}_bytes_total"]) for x in rows)
    dom = [x for x in rows if x["kernel"].startswith("gemm_glds_kernel<") and int(x["kernel"].split("<")[1].split(",")[0]) >= 192]
    return tot, sum(float(x[f"{c}_bytes_total"]) for x in dom), sum(int(x["launches"]) for x in dom)
ft, fd, n = pmc("FETCH_SIZE"); wt, wd, _ = pmc("WRITE_SIZE")
steps_in_pmc = 4.0 + d['config'].get('settle_steps', 0)      # conditioning + 1 warm-up + 2 timed + 1 roofline-leg step in the profiled command (+ the final decode)
traffic = (f"{(ft + wt) / steps_in_pmc / 1e9:.0f} GB per step in total (≈ {(ft + wt) / steps_in_pmc / 1e9 / d['ms_per_step']:.1f} TB/s average); the dominant kernel "
           f"moves {(fd + wd) / n / 1e6:.0f} MB per launch against {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic (operands once + outputs), i.e. "
           f"{(fd + wd) / n / r['algorithmic_bytes_per_launch']:.2f}×")
f = lambda v: f"{v:.1e}".replace("e-0", "e-") if v else "0"
mf = {x["kernel"]: x for x in csv.DictReader(open(f"{P}/r03_pmc_mfma_util_per_kernel.csv"))} if os.path.exists(f"{P}/r03_pmc_mfma_util_per_kernel.csv") else {}
dom_mf = [x for k, x in mf.items() if k.startswith("gemm_glds_kernel<") and int(k.split("<")[1].split(",")[0]) >= 192]
mfma_util = (sum(float(x["SQ_VALU_MFMA_BUSY_CYCLES_total"]) for x in dom_mf) / (4.0 * sum(float(x["SQ_BUSY_CU_CYCLES_total"]) for x in dom_mf))) if dom_mf else 0.0
rep = {"VALUE": f"{d['value']:.2f}", "MS": f"{d['ms_per_step']:.1f}", "CLIPS": f"{d['config']['clip_steps_per_sec']:.0f}", "RATIO": f"{cb['gpu_over_cpu']:.0f}",
       "CPUMODEL": cb.get("cpu_model", "host CPU"), "CPU1": f"{cb['batch1_seconds_per_step']:.2f}", "CPU8": f"{cb['batch8_seconds_per_step']:.1f}", "CPUVAL": f"{cb['value']:.3f}",
       "STAGES": stages, "DOMMS": f"{r['kernel_ms_per_step']:.1f}", "DOMTF": f"{r['achieved']:.0f}", "DOMFRAC": f"{100 * r['frac']:.1f} %",
       "NLAUNCH": str(r['all_gemm_kernels']['launches_per_step']), "DOMN": str(r['launches_per_step']), "DOMALGO": f"{r['algorithmic_tflop_per_step']:.1f}",
       "MFMAUTIL": f"{100 * mfma_util:.0f} %",
       "ALLTF": f"{r['all_gemm_kernels']['achieved']:.0f}", "STEPTF": f"{r['whole_step']['achieved']:.0f}", "TRAFFIC": traffic,
       "MELMS": f"{d['mel_path']['stage_ms']:.2f}", "MELGB": f"{d['mel_path']['achieved_GBps']:.0f}", "MELFRAC": f"{100 * d['mel_path']['frac']:.1f} %",
       "W3": f"{w['dsg_phase_audioldm2']['value']:.1f}", "W4": f"{w['mpgd_sr4']['value']:.1f}", "W5": f"{w['diffmusic_style_audioldm2']['value']:.1f}",
       "FDEC": f"{d['after_loop']['final_decode_ms']:.0f}",
       "CPUHOST": str(cb.get("host_logical_cpus")), "CPUQUOTA": f"{cb.get('cgroup_cpu_quota') or 0:.0f}", "CPUTHREADS": str(cb.get("threads_used")),
       "RATIO1": f"{d['config']['clip_steps_per_sec'] / 8 / cb['batch1_x8_steps_per_sec']:.0f}",
       "ISSUED": f"{r['all_gemm_kernels']['issued_tflop_per_step']:.1f}",
       "S3": f"{s3['value']:.2f}", "S3C": f"{s3['config']['clip_steps_per_sec']:.0f}", "S4": f"{s4['value']:.2f}", "S4C": f"{s4['config']['clip_steps_per_sec']:.0f}"}
for tag, name in (("P2", "dps_inpainting"), ("P3", "dsg_phase_audioldm2"), ("P4", "mpgd_sr4"), ("P5", "diffmusic_style_audioldm2")):
    q = json.load(open(f"{P}/r03_fullsize_parity_{name}.json"))
    rep.update({tag + "EPS": f(q["unet_eps"]), tag + "VAE": f"{f(q['vae_mel'])} / {f(q['vae_bwd'])}",
                tag + "VOC": f"{f(q['vocoder_wav'])} / {f(q['vocoder_bwd'])} ({q['vocoder_bwd_cos']:.4f})",
                tag + "OP": f"{f(q['operator_loss'])} / {f(q['operator_bwd'])}",
                tag + "STEP": f"{f(q['step_loss'])} / {f(q['step_grad'])} ({q['step_grad_cos']:.4f}) / {f(q['step_prev_sample'])}"})
s = open("DESIGN.template.md").read()
for k, v in rep.items():
    s = s.replace(f"@@{k}@@", v)
import re
left = re.findall(r"@@\w+@@", s)
assert not left, left
open("DESIGN.md", "w").write(s)
print("DESIGN.md written;", rep["VALUE"], "steps/s")
