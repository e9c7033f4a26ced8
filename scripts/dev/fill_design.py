"""Fills the @@...@@ placeholders of DESIGN.template.md from the committed evidence (profiles/r02_*) -> DESIGN.md."""
import csv, json, os
P = "profiles"
d = json.load(open(f"{P}/r02_bench.json"))
r, cb = d["roofline"], d["cpu_baseline"]
w = {k: json.load(open(f"{P}/r02_bench_{k}.json")) for k in ("dsg_phase_audioldm2", "mpgd_sr4", "diffmusic_style_audioldm2")}
st = d["stage_ms"]
stages = (f"U-Net(2B)+CFG {st['unet_cfg']:.1f}, VAE fwd {st['vae_fwd']:.1f} / bwd {st['vae_bwd']:.1f}, HiFi-GAN fwd {st['hifigan_fwd']:.1f} / bwd "
          f"{st['hifigan_bwd']:.1f}, operator+mel+loss fwd+bwd {st['operator_mel_loss_fwd_bwd']:.2f}, update {st['sched_update']:.3f}")
def pmc(c):
    rows = [x for x in csv.DictReader(open(f"{P}/r02_pmc_{c}_per_kernel.csv"))]
    tot = sum(float(x[f"{c}_bytes_total"]) for x in rows)
    dom = [x for x in rows if x["kernel"].startswith("gemm_glds_kernel<") and int(x["kernel"].split("<")[1].split(",")[0]) >= 192]
    return tot, sum(float(x[f"{c}_bytes_total"]) for x in dom), sum(int(x["launches"]) for x in dom)
ft, fd, n = pmc("FETCH_SIZE"); wt, wd, _ = pmc("WRITE_SIZE")
steps_in_pmc = 4.0            # 1 warm-up + 2 timed + 1 roofline-leg step in the profiled command (+ the final decode)
traffic = (f"{(ft + wt) / steps_in_pmc / 1e9:.0f} GB per step in total (≈ {(ft + wt) / steps_in_pmc / 1e9 / d['ms_per_step']:.1f} TB/s average); the dominant kernel "
           f"moves {(fd + wd) / n / 1e6:.0f} MB per launch against {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic (operands once + outputs), i.e. "
           f"{(fd + wd) / n / r['algorithmic_bytes_per_launch']:.2f}×")
p5 = json.load(open(f"{P}/r02_fullsize_parity_diffmusic_style_audioldm2.json")) if os.path.exists(f"{P}/r02_fullsize_parity_diffmusic_style_audioldm2.json") else None
f = lambda v: f"{v:.1e}".replace("e-0", "e-")
rep = {"VALUE": f"{d['value']:.2f}", "MS": f"{d['ms_per_step']:.1f}", "CLIPS": f"{d['config']['clip_steps_per_sec']:.0f}", "RATIO": f"{cb['gpu_over_cpu']:.0f}",
       "STAGES": stages, "DOMMS": f"{r['kernel_ms_per_step']:.1f}", "DOMTF": f"{r['achieved']:.0f}", "DOMFRAC": f"{100 * r['frac']:.1f} %",
       "ALLTF": f"{r['all_gemm_kernels']['achieved']:.0f}", "STEPTF": f"{r['whole_step']['achieved']:.0f}", "TRAFFIC": traffic,
       "MELMS": f"{d['mel_path']['stage_ms']:.2f}", "MELGB": f"{d['mel_path']['achieved_GBps']:.0f}", "MELFRAC": f"{100 * d['mel_path']['frac']:.1f} %",
       "W3": f"{w['dsg_phase_audioldm2']['value']:.1f}", "W4": f"{w['mpgd_sr4']['value']:.1f}", "W5": f"{w['diffmusic_style_audioldm2']['value']:.1f}",
       "FDEC": f"{d['after_loop']['final_decode_ms']:.0f}"}
if p5:
    rep.update({"P5EPS": f(p5["unet_eps"]), "P5VAE": f"{f(p5['vae_mel'])} / {f(p5['vae_bwd'])}",
                "P5VOC": f"{f(p5['vocoder_wav'])} / {f(p5['vocoder_bwd'])} ({p5['vocoder_bwd_cos']:.4f})",
                "P5OP": f"{f(p5['operator_loss'])} / {f(p5['operator_bwd'])}",
                "P5STEP": f"{f(p5['step_loss'])} / {f(p5['step_grad'])} ({p5['step_grad_cos']:.4f}) / {f(p5['step_prev_sample'])}"})
s = open("DESIGN.template.md").read()
for k, v in rep.items():
    s = s.replace(f"@@{k}@@", v)
import re
left = re.findall(r"@@\w+@@", s)
assert not left, left
open("DESIGN.md", "w").write(s)
print("DESIGN.md written;", rep["VALUE"], "steps/s")
