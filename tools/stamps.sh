# usage (on the GPU box): bash tools/stamps.sh  -- share of wavefront cycles per phase of glfgen_kernel (diagnostics build)
make -s -C bcftools_amd/csrc clean >/dev/null; make -s -j8 -C bcftools_amd/csrc DIAG=1 >/dev/null 2>&1 || { echo "diag build failed"; exit 1; }
BCFGPU_STAMPS=1 python bench.py --sites 16384 --steps 4 --warmup 1 --cpu-seconds 0 --cpu-all-cores 0 --extras 0 2>&1 | grep "glfgen stamps" | tail -2
make -s -C bcftools_amd/csrc clean >/dev/null; make -s -j8 -C bcftools_amd/csrc >/dev/null 2>&1
