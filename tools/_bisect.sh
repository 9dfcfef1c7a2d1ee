mkdir -p gpurun_out/r2b
for f in test_abi test_backbone_gpu test_e2e_gpu test_fullsize_gpu test_known_answers test_ops_gpu test_parity_r2_gpu; do
  python -m pytest tests/$f.py "tests/test_train_step_gpu.py::test_two_training_iterations_and_resume[mse]" "tests/test_parity_r2_gpu.py::test_real_world_through_keymorph_forward" -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r2b/$f.log
  echo "$f: $(tail -1 gpurun_out/r2b/$f.log)"
done
