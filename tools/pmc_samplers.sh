# SQ counters of the rejection-sampler kernels (lane utilisation of the Marsaglia-Tsang loop): bash tools/pmc_samplers.sh  (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_samplers; rm -rf "$OUT"; mkdir -p "$OUT"
for m in beta_bernoulli gamma_normal; do
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq_$m" -- python3 tools/time_samplers.py --no-cpu $m > "$OUT/sq_$m.log" 2>&1 || { tail -5 "$OUT/sq_$m.log"; exit 1; }
echo "== $m"; python3 tools/pmc_by_kernel.py "$OUT/sq_$m" | grep -A1 "gjx_plan_kernel" | head -8
done
find "$OUT" -name "*.csv" -size +1M -delete
