#!/bin/bash
# one gpurun call = several measurements; everything lands in gpurun_out/$TAG/ (scratch: copy what is to be judged into profiles/)
TAG=${1:-call}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*" | tee -a $OUT/index.txt; ( time timeout ${TMO:-600} "$@" ) > $OUT/$name.log 2>&1; echo "    rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)" | tee -a $OUT/index.txt; }
for step in "$@"; do
case $step in
  parity) GO1_PARITY_LOG=$PWD/$OUT/parity_rates.txt run parity python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py -q -s -k "product_instances or train_eval_split or deferred_torque" ;;
  altseed) GO1_PARITY_ALT=1 GO1_PARITY_LOG=$PWD/$OUT/parity_rates_alt_seed.txt run altseed python -m pytest tests/test_gpu_parity.py -q -s -k "product_instances" ;;
  pmc) GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} run pmc bash tools/pmc.sh $TAG ;;
  quick) GO1_PARITY_LOG=$PWD/$OUT/parity_rates.txt run quick python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py tests/test_gpu_ppo_reference.py -q -k "product_instances or failed or full_step_matches or ragged or deferred_torque or gpu_fp32_update or history" ;;
  fusedtests) run fusedtests python -m pytest tests/test_gpu_ppo_fused.py -q -x ;;
  gemm) run gemm python tools/bench_gemm.py ;;
  bench) run bench python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --breakdown ;;
  bench_atomics) GO1_WGRAD_SLABS=0 run bench_atomics python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --breakdown ;;
  disttests) run disttests python -m pytest tests/test_gpu_distributed.py -q -x ;;
  benchfull) TMO=900 run benchfull python bench.py ;;
  prof) (cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -- python $OLDPWD/bench.py --steps 3 --warmup 2 --headline-only --no-cpu-baseline) > $OUT/prof.log 2>&1; find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; ; for t in $(find $OUT/prof -name "*kernel_trace.csv"); do python tools/timeline.py $t --which -3 > $OUT/timeline.txt 2>&1; python tools/timeline.py $t --anchor mse_kernel --which -3 >> $OUT/timeline.txt 2>&1; python tools/timeline.py $t --anchor go1_step_kernel --which -5 > $OUT/timeline_rollout.txt 2>&1; done; find $OUT/prof -type f ! -name "*stats*" -delete; echo "=== prof rc=$?" | tee -a $OUT/index.txt ;;
  changed) GO1_PARITY_LOG=$PWD/$OUT/parity_rates.txt TMO=330 run changed python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py -x -q --durations=15 -k "not (learns or play_eval or teacher or unchanged_train or graph_replay or autograd_update or two_rank or rccl_one_rank or training_survives)" ;;
  rough) GO1_PARITY_LOG=$PWD/$OUT/parity_rates.txt TMO=200 run rough python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py -x -q --durations=8 -k "height_field or (product_instances and not plane) or rough_terrain_env_end_to_end or height_scan" ;;
  twin) run twin python tests/twin_probe.py ;;
  traj) run traj python -m pytest tests/test_gpu_ppo_fused.py -q -s -k "tracks_autograd" ;;
  rest) TMO=150 run rest_other python -m pytest tests -m gpu -q --durations=10 --ignore=tests/test_gpu_parity.py --ignore=tests/test_gpu_env.py
        TMO=200 run rest_env python -m pytest tests/test_gpu_env.py -m gpu -q --durations=10 -k "learns or play_eval or teacher or unchanged_train or graph_replay or autograd_update or two_rank or rccl_one_rank or training_survives" ;;
  pmc_walls) GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} PMC_EXTRA=--rough run pmc_walls bash tools/pmc.sh ${TAG}_walls ;;
  gputests) GO1_PARITY_LOG=$PWD/$OUT/parity_rates.txt TMO=1500 run gputests python -m pytest tests/ -x -q -m gpu --durations=30 ;;
  dropin) run dropin python -m pytest tests/test_gpu_env.py -q -x -s -k "unchanged_train_script" ;;
  ab) run ab python tools/probes/step_variant_ab.py $(ls walk-these-ways_amd/csrc/variants/*.so | grep -v prof) ;;
  ab_resets) AB_RESETS=1 run ab_resets python tools/probes/step_variant_ab.py $(ls walk-these-ways_amd/csrc/variants/*.so | grep -v "prof\|r04") ;;
  ab_rough_resets) AB_RESETS=1 AB_ROUGH=1 AB_REPS=3 run ab_rough_resets python tools/probes/step_variant_ab.py $(ls walk-these-ways_amd/csrc/variants/*.so | grep -v "prof\|r04") ;;
  ab_rough) AB_ROUGH=1 AB_REPS=3 run ab_rough python tools/probes/step_variant_ab.py $(ls walk-these-ways_amd/csrc/variants/*.so | grep -v "prof\|r04") ;;
  icache) run icache tools/probes/icache_probe ;;
  phases_rough) run phases_rough python tools/phase_profile.py --lib walk-these-ways_amd/csrc/variants/prof.so --steps 32 --rough ;;
  sweep) AB_ENVS=64,256,1024,2048,4096,8192 run sweep python tools/probes/step_variant_ab.py ;;
  phases) run phases python tools/phase_profile.py --lib walk-these-ways_amd/csrc/variants/prof.so --steps 32; run phases_standing python tools/phase_profile.py --lib walk-these-ways_amd/csrc/variants/prof.so --steps 32 --zero-actions ;;
  *) echo "unknown step $step" ;;
esac
done
cat $OUT/index.txt
