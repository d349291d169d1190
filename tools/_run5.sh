mkdir -p gpurun_out/r2e
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --ablate 8192 > /dev/null 2> gpurun_out/r2e/vga.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --ablate 8192 --workload 1280x960 --batch 64 > /dev/null 2> gpurun_out/r2e/w1280.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --ablate 8192 --workload 1280x960 --batch 64 --wgs-per-cu 4 > /dev/null 2> gpurun_out/r2e/w1280_4.err
for f in vga w1280 w1280_4; do echo $f; grep "cycles/strip" gpurun_out/r2e/$f.err | head -2 | tail -1; done
bash tools/pmc_kernel.sh r2e_pmc_vga
bash tools/pmc_kernel.sh r2e_pmc_1280 --workload 1280x960 --batch 64
python - <<'P'
import csv,collections
for n in ('r2e_pmc_vga','r2e_pmc_1280'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f'gpurun_out/{n}/p_counter_collection.csv')):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1].split('<')[0]
        if k not in ('k_fused_strips','k_gather_orb'): continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
    for k,d in agg.items():
        print(n,k,{c: round(v/cnt[k][c]/1e6,2) for c,v in d.items()})
P
