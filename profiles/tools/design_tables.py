"""Regenerate the two result tables of DESIGN.md (section 5 kernel table, section 7 config table) from the round's profile files:
profiles/<TAG>_cfgN_kernel_stats.csv, profiles/pmc_traffic.json, profiles/<TAG>_bench_cfgN.json.
    python3 profiles/tools/design_tables.py [TAG]        (default r06)
The tables sit between the markers <!-- kernel-table --> / <!-- /kernel-table --> and <!-- results-table --> / <!-- /results-table -->
(a fresh DESIGN.md carries @@KERNEL_TABLE@@ / @@RESULTS_TABLE@@ instead)."""
import csv
import json
import os
import re
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
p = os.path.join(R, "DESIGN.md")
s = open(p).read()
t = json.load(open(os.path.join(R, "profiles", "pmc_traffic.json")))
CFGS = ("cfg2", "cfg3", "cfg4", "cfg5")
LB = {"cfg2": 13, "cfg3": 13, "cfg4": 13, "cfg5": 14}
STAGES = (("forward_spectra", "al_forward_spectra"), ("spectral_mac", "al_spectral_mac"), ("block_synthesis", "al_block_synthesis"),
          ("k_mixdown", "al_mixdown"), ("emitter_gains", "al_emitter_gains"), ("event_levels", "al_event_levels"))


def kms(c):
    out = {}
    for r in csv.DictReader(open(os.path.join(R, "profiles", f"{TAG}_{c}_kernel_stats.csv"))):
        for key, stage in STAGES:
            if key in r["Name"]:
                out[stage] = out.get(stage, 0) + float(r["AverageNs"]) / 1e6
    return out


d = {c: json.loads(open(os.path.join(R, "profiles", f"{TAG}_bench_{c}.json")).read().strip().splitlines()[-1]) for c in CFGS}
d20 = json.loads(open(os.path.join(R, "profiles", f"{TAG}_bench_cfg2_steps20.json")).read().strip().splitlines()[-1])
k = {c: kms(c) for c in CFGS}
traffic = {c: t[f"{c}/log2_block={LB[c]}"] for c in CFGS}
valu = {c: t.get(f"{c}/log2_block={LB[c]}/valu_insts", {}) for c in CFGS}
tot = {c: sum(traffic[c].values()) / 1e9 for c in CFGS}
fmt = lambda v: format(int(round(v)), ",").replace(",", " ")   # noqa: E731
ISSUE = 256 * 4 * 2.4e9 / 2.0


def cell(c, stage):
    ms, gb = k[c][stage], traffic[c][stage] / 1e9
    iss = valu[c].get(stage, 0) / ISSUE / (ms * 1e-3) if valu[c].get(stage) else None
    gf = d[c]["roofline"].get("gflops_by_stage", {}).get(stage)
    txt = (f"{ms:.3f} ms, {gb:.2f} GB, {gb / ms:.1f} TB/s" if ms < 1 else f"{ms:.2f} ms, {gb:.1f} GB, {gb / ms:.1f} TB/s")
    if gf:
        txt += f"; {gf / 1e3:.1f} TFLOP/s"
    if iss:
        txt += f", issue {iss:.2f}"
    dom = max((st for _, st in STAGES[:4]), key=lambda st: k[c][st])
    return f"**{txt}**" if stage == dom else txt


names = {"al_forward_spectra": "forward transforms (IR partitions + signal windows): `k_forward_spectra_split<13>`; cfg5 `k_forward_spectra_quad16`",
         "al_spectral_mac": "accumulate: `k_spectral_mac_static<12,P,NKTW>` (cfg2 P = 12, two k-tiles; cfg4 P = 6; cfg5 P = 12 at B = 16384), `k_spectral_mac_moving<6,12,1>` (cfg3)",
         "al_block_synthesis": "inverse transforms + overlap-save + level partials: `k_block_synthesis_split<13>`; cfg5 `k_block_synthesis_quad16`",
         "al_mixdown": "`k_mixdown`"}
rows = ["| Kernel (default dispatch): ms, PMC bytes, rate on them; plan-counted TFLOP/s, VALU issue fraction | cfg2 | cfg3 | cfg4 | cfg5 (B = 16384) |",
        "|---|---|---|---|---|"]
for stage, label in names.items():
    rows.append(f"| {label} | " + " | ".join(cell(c, stage) for c in CFGS) + " |")
rows.append("| `k_emitter_gains` + `k_event_levels` | " + " | ".join(f"{k[c]['al_emitter_gains'] + k[c]['al_event_levels']:.3f} ms" for c in CFGS) + " |")
rows.append("| scene: bench `ms_per_step` (median of 3 × K steps; the pool's boxes differ by about 3 %), PMC bytes | "
            + " | ".join(f"**{d[c]['ms_per_step']:.2f} ms**, {tot[c]:.1f} GB" for c in CFGS) + " |")
alg = {c: d[c]["roofline"]["algorithmic_bytes_per_launch"] / 1e9 for c in CFGS}
rows.append("| algorithmic bytes, scene-only contract (SURVEY 8d: inputs + `scene.audio`); `roofline.frac` / `path_frac`; `traffic_ratio` | "
            + " | ".join(f"{alg[c]:.3f} GB; {d[c]['roofline']['frac']:.3f} / {d[c]['roofline']['path_frac']:.3f}; {tot[c] / alg[c]:.1f}×" for c in CFGS) + " |")
full = {c: d[c]["roofline"].get("contracts", {}).get("full_api") for c in CFGS}
if all(full.values()):
    rows.append("| ... full-API contract (+ every `event.spatial_audio` written once); `frac` / `path_frac`; `traffic_ratio` | "
                + " | ".join(f"{full[c]['algorithmic_bytes'] / 1e9:.3f} GB; {full[c]['frac']:.3f} / {full[c]['path_frac']:.3f}; "
                             f"{tot[c] / (full[c]['algorithmic_bytes'] / 1e9):.1f}×" for c in CFGS) + " |")
kernel_table = "<!-- kernel-table -->\n" + "\n".join(rows) + "\n<!-- /kernel-table -->"


def parity_of(c):
    par = d[c].get("parity")
    if not par:
        return "—"
    full = par.get("events_in_full", par["events"])
    txt = f"{par['rel_rms']:.1e} / {par['max_abs_over_peak']:.1e} ({full} event{'s' if full != 1 else ''} in full, {par['rows']} × {fmt(par['samples'])}"
    rs = par.get("rows_sampled")
    if rs:
        txt += (f"; one row of each of the other {rs['events']}: worst {rs['rel_rms_worst_row']:.1e} / {rs['max_abs_over_peak_worst_row']:.1e}, "
                f"level invariant within {rs['level_invariant_worst_rel_err']:.0e}")
    return txt + ")"


def cpu_of(c):
    one = d[c].get("cpu_baseline", {})
    allc = d[c].get("cpu_baseline_all_cores")
    txt = f"{one.get('value', float('nan')):.2f}" + (" (extrapolated)" if one.get("extrapolated") else "")
    if allc and "value" in allc:
        txt += f" / {allc['value']:.1f} on {allc['cores']} cores"
    return txt


res = ["| Config | ms per scene | scene-s/s | dominant kernel | `frac` / `path_frac` | GFLOP/s (dominant), fraction of 157.3 TF | parity vs oracle: rel. RMS / max-abs | oracle: 1 core / all allowed cores |",
       "|---|---|---|---|---|---|---|---|"]
label = {"cfg2": "cfg2 (headline)", "cfg3": "cfg3 (16 moving events × 32 IRs)", "cfg4": "cfg4 (30 s scene, 32 events, 1 s RIR)",
         "cfg5": "cfg5 (64 capsules, 128 events, 4 s RIR, ambience, folded FX)"}
for c in CFGS:
    r = d[c]["roofline"]
    ms = f"{d[c]['ms_per_step']:.2f}" + (f" ({d20['ms_per_step']:.2f} with the driver's `--steps 20`)" if c == "cfg2" else "")
    res.append(f"| {label[c]} | {ms} | **{fmt(d[c]['value'])}** | `{r['kernel']}` {r['kernel_ms'][r['kernel']]:.3f} ms | {r['frac']:.3f} / {r['path_frac']:.3f} | "
               f"{fmt(r['gflops'])}, {r['valu_frac_of_peak']:.2f} | {parity_of(c)} | {cpu_of(c)} |")
e2e, drop = d["cfg2"].get("end_to_end"), d["cfg2"].get("end_to_end_dropin")
tail = ""
if e2e and drop:
    tail = (f"\n\nPCIe-inclusive on the same run (never the headline): pipelined batch driver **{fmt(e2e['value'])} scene-s/s** "
            f"({e2e['ms_per_scene']:.1f} ms per cfg2 scene), synchronous drop-in `Scene.generate()` **{fmt(drop['value'])}** ({drop['ms_per_scene']:.1f} ms).")
results_table = (f"<!-- results-table -->\nRound-{int(TAG[1:3])} results (`profiles/{TAG}_bench_*.json`: one MI355X, host {d['cfg2']['cpu_baseline']['cpu_model']}, final sources "
                 f"`{d['cfg2']['config']['source_hash']}`; kernel rows in section 5):\n\n" + "\n".join(res) + tail + "\n<!-- /results-table -->")

for marker, block, tag in (("@@KERNEL_TABLE@@", kernel_table, "kernel-table"), ("@@RESULTS_TABLE@@", results_table, "results-table")):
    if marker in s:
        s = s.replace(marker, block)
    else:
        s = re.sub(rf"<!-- {tag} -->.*?<!-- /{tag} -->", lambda m, b=block: b, s, flags=re.S)
open(p, "w").write(s)
print("DESIGN.md tables regenerated from", TAG)
