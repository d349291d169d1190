mkdir -p gpurun_out/r5g
python - <<P
import numpy as np
z = np.load("tests/golden/demo_pyramid.npz")
z["img"].astype(np.uint8).tofile("/tmp/demo_pyramid.raw")
P
make -s -C tools pislam_demo > /dev/null 2>&1
for s in 1 3; do for f in 1 0; do for g in 1 0; do
  echo "streams $s frame $f graphs $g: $(tools/pislam_demo /tmp/demo_pyramid.raw --batch 1 --steps 3000 --streams $s --opt frame=$f --opt graphs=$g 2>&1 | head -1)" | tee -a gpurun_out/r5g/cpp_one.txt
done; done; done
for f in 1 0; do
  echo "batch 2 streams 1 frame $f: $(tools/pislam_demo /tmp/demo_pyramid.raw --batch 2 --steps 3000 --streams 1 --opt frame=$f 2>&1 | head -1)" | tee -a gpurun_out/r5g/cpp_one.txt
done
