cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu -k "training_step or multi_agent or train" 2>&1 | tail -3
for rep in 1 2; do
for l in obs_nolist liboc_amd; do
  echo "== $l"
  for lay in cramped_room asymmetric_advantages; do
    OC_AMD_LIB=$PWD/overcooked_ai_amd/$l.so timeout 100 python tools/time_train_step.py $lay 65536 2>&1 | grep "use_phi=True"
  done
done
done
