#!/bin/bash
# One gpurun call = several measurements. Usage: scripts/gpu_session.sh <tag> <step> [<step> ...]
# Steps: tests, ordered, ahead, tests_new, fpslab, bw, bq, models, prof_model(s), bench, smoke, trainbench, trainstep, profile, ...
# (the case list below is the reference). Logs land in gpurun_out/<tag>/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for step in "$@"; do
    echo "=== $step $(date +%T)"
    case $step in
    tests)     timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; tail -5 "$OUT/tests.log" ;;
    ordered)   timeout 600 python -m pytest tests/test_fps_ordered_gpu.py tests/test_modules_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > "$OUT/ordered.log" 2>&1; tail -15 "$OUT/ordered.log" ;;
    ahead)     timeout 600 python -m pytest tests/test_geometry_ahead_gpu.py tests/test_modules_gpu.py -m gpu -q > "$OUT/ahead.log" 2>&1; tail -25 "$OUT/ahead.log" ;;
    tests_new) timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_ball_cells_gpu.py -m gpu -q > "$OUT/tests_new.log" 2>&1; tail -15 "$OUT/tests_new.log" ;;
    fpslab)    for f in build_lab/fps_*; do n=$(basename $f); case $n in *lab*|*prof*) continue;; esac; timeout 90 $f $n > "$OUT/$n.log" 2>&1; grep "n= 4096" "$OUT/$n.log"; done ;;
    prof_bw)   (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bw" -- python $ROOT/scripts/bw_probe.py > "$OUT/prof_bw.log" 2>&1); find "$OUT/prof_bw" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_bw_probe.csv" \;; rm -rf "$OUT/prof_bw"; cut -d, -f1-8 "$OUT/kernel_stats_bw_probe.csv" | head -30 ;;
    prof_bq)   (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bq" -- python $ROOT/scripts/bq_probe.py msg > "$OUT/prof_bq.log" 2>&1); find "$OUT/prof_bq" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_bq_msg.csv" \;; rm -rf "$OUT/prof_bq"; cut -d, -f1-8 "$OUT/kernel_stats_bq_msg.csv" | head -30 ;;
    bw)        timeout 300 python scripts/bw_probe.py > "$OUT/bw_probe.log" 2>&1; cat "$OUT/bw_probe.log" ;;
    bq)        timeout 300 python scripts/bq_probe.py > "$OUT/bq_probe.log" 2>&1; cat "$OUT/bq_probe.log" ;;
    tests_fp)  timeout 600 python -m pytest tests/test_fp_mlp_gpu.py tests/test_sa_mlp_gpu.py tests/test_modules_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > "$OUT/tests_fp.log" 2>&1; tail -15 "$OUT/tests_fp.log" ;;
    models)    timeout 600 python scripts/model_forward_bench.py > "$OUT/model_forward.log" 2>&1; cat "$OUT/model_forward.log" ;;
    prof_model) (cd /tmp && export TMPDIR=/tmp && PN2_MODEL_FUSED_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_model" -- python $ROOT/scripts/model_forward_bench.py ${MODEL:-sem_seg} > "$OUT/prof_model.log" 2>&1); find "$OUT/prof_model" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_model_${MODEL:-sem_seg}.csv" \;; rm -rf "$OUT/prof_model"; head -30 "$OUT/kernel_stats_model_${MODEL:-sem_seg}.csv" | cut -c1-150 ;;
    prof_models) for MODEL in ${MODELS:-cls_ssg cls_msg part_seg sem_seg}; do (cd /tmp && export TMPDIR=/tmp && PN2_MODEL_FUSED_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_model" -- python $ROOT/scripts/model_forward_bench.py $MODEL > "$OUT/prof_model_$MODEL.log" 2>&1); find "$OUT/prof_model" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_model_$MODEL.csv" \;; rm -rf "$OUT/prof_model"; grep fused "$OUT/prof_model_$MODEL.log"; head -14 "$OUT/kernel_stats_model_$MODEL.csv" | cut -c1-140; done ;;
    mlplab)    for f in "" $(ls build_lab/libpn2ops_*.so); do echo "--- ${f:-product}"; PN2OPS_LIBRARY=${f:+$ROOT/$f} PN2_MLP_BENCH_KERNEL_ONLY=1 timeout 200 python scripts/sa_mlp_bench.py 2>&1 | grep kernel | grep -E "${MLPLAB_FILTER:-.}"; done > "$OUT/mlplab.log" 2>&1; cat "$OUT/mlplab.log" ;;
    mlpacc)    timeout 300 python scripts/mlp_accuracy.py > "$OUT/mlp_accuracy.log" 2>&1; cat "$OUT/mlp_accuracy.log" ;;
    mlpbench)  timeout 300 python scripts/sa_mlp_bench.py --json "$OUT/sa_mlp_bench.json" > "$OUT/sa_mlp_bench.log" 2>&1; cut -c1-200 "$OUT/sa_mlp_bench.log" ;;
    tests_mlp) timeout 600 python -m pytest tests/test_sa_mlp_gpu.py tests/test_fp_mlp_gpu.py -m gpu -q -x > "$OUT/tests_mlp.log" 2>&1; tail -15 "$OUT/tests_mlp.log" ;;
    smoke)     timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" ;;
    configs)   timeout 600 python scripts/config_shapes.py > "$OUT/config_shapes.json" 2> "$OUT/config_shapes.err"; tail -3 "$OUT/config_shapes.json" ;;
    bench)     timeout 600 python bench.py > "$OUT/bench.log" 2>&1; tail -3 "$OUT/bench.log" ;;
    trainbench) timeout 400 python scripts/train_mlp_bench.py 2> "$OUT/train_mlp_bench.err" | grep "^{" > "$OUT/train_mlp_bench.jsonl"; python - "$OUT/train_mlp_bench.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print("%-58s fused %7.1f + %7.1f us | layer by layer %7.1f + %7.1f us | x%.2f" % (r["level"][:58], r["fused"]["forward_us"], r["fused"]["backward_us"], r["layer_by_layer"]["forward_us"], r["layer_by_layer"]["backward_us"], r["speedup_step"]))
PY
    ;;
    trainstep) timeout 900 python scripts/train_step_bench.py --steps 8 --graph 2> "$OUT/train_step.err" | grep "^{" > "$OUT/train_step.jsonl"; python - "$OUT/train_step.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    f, u = r["fused"], r["layer_by_layer"]
    d = r.get("fused_direct", f)
    g, ga = r.get("fused_direct_graph", {}), r.get("fused_direct_graph_ahead", {})
    print("%-50s layer by layer %6.2f + %6.2f = %6.2f ms | fused %6.2f + %6.2f = %6.2f ms x%.2f | gradients added in the kernels %6.2f ms x%.2f | as one HIP graph %s ms, with the geometry one step ahead %s ms" % (r["model"][:50], u["forward_ms"], u["backward_ms"], u["step_ms"], f["forward_ms"], f["backward_ms"], f["step_ms"], r["speedup"], d["step_ms"], r.get("speedup_direct", 0.0), g.get("step_ms", "-"), ga.get("step_ms", "-")))
PY
    ;;
    trainrccl) timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 scripts/train_step_bench.py sem_seg --steps 8 --fused-only 2> "$OUT/train_rccl.err" | grep "^{" > "$OUT/train_step_rccl.jsonl"; cat "$OUT/train_step_rccl.jsonl" | cut -c1-400
               timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --no-extras 2> "$OUT/bench_rccl.err" | grep "^{" > "$OUT/bench_rccl.json"; python -c "import json,sys; print(json.load(open(sys.argv[1]))['allreduce'])" "$OUT/bench_rccl.json" ;;
    traincheck) PN2_TRAIN_OPTS=top_stored=0 timeout 300 python scripts/train_mlp_check.py > "$OUT/train_check_zfree.log" 2>&1; tail -1 "$OUT/train_check_zfree.log"; timeout 300 python scripts/train_mlp_check.py > "$OUT/train_check.log" 2>&1; tail -1 "$OUT/train_check.log" ;;
    traintests) timeout 1500 python -m pytest tests/test_train_mlp_gpu.py tests/test_train_golden_gpu.py tests/test_train_fuzz_gpu.py tests/test_whole_model_fp64_gpu.py tests/test_overlap_status_gpu.py tests/test_modules_gpu.py -m gpu -q > "$OUT/traintests.log" 2>&1; tail -25 "$OUT/traintests.log" ;;
    wholemodel) timeout 900 python scripts/whole_model_fp64.py --top 2> "$OUT/whole_model_fp64.err" | grep "^{" > "$OUT/whole_model_fp64.jsonl"; python - "$OUT/whole_model_fp64.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print("%-52s grad L2 err vs float64: layer by layer %.2e  fused %.2e  (fused vs layer by layer %.2e)" % (r["model"][:52], r["layer_by_layer"]["grad_l2_err"], r["fused"]["grad_l2_err"], r["fused_vs_layer_by_layer_l2"]))
PY
    ;;
    fuzzsurvey) timeout ${FUZZ_TIMEOUT:-900} python scripts/train_fuzz_survey.py ${FUZZ_CASES:-600} > "$OUT/train_fuzz_survey.txt" 2> "$OUT/train_fuzz_survey.err"; tail -5 "$OUT/train_fuzz_survey.txt" ;;
    timeline)  for lv in ${TL_LEVELS:-sem_seg:SA4 FP:sem_seg}; do name=$(echo $lv | tr ':' ' '); (cd /tmp && export TMPDIR=/tmp && PN2_TRAIN_OPTS="${PN2_TRAIN_OPTS:-}" PN2_TRAIN_BENCH_KERNEL_ONLY=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl_$lv" -- python $ROOT/scripts/train_mlp_bench.py "$name" > "$OUT/tl_$lv.log" 2>&1); f=$(find "$OUT/tl_$lv" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/level_timeline.py "$f" "$OUT/timeline_$lv.txt"; rm -rf "$OUT/tl_$lv"; cat "$OUT/timeline_$lv.txt"; done ;;
    profile)   timeout 1200 bash scripts/profile_round.sh > "$OUT/profile.log" 2>&1; tail -5 "$OUT/profile.log" ;;
    *)         echo "unknown step $step" ;;
    esac
done
echo "=== done $(date +%T)"
