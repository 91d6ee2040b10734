B=behavior_driven_video_synthesis_amd/build
for rep in 1 2; do
for t in ${VARIANTS:-default}; do
  if [ $t = default ]; then unset VUNET_HIP_LIB; else export VUNET_HIP_LIB=$PWD/$B/libvunet_hip_$t.so; fi
  f=$(python tools/time_seq_train.py --stage flow 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['step_ms'])")
  c=$(python tools/time_seq_train.py --stage cvae 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['step_ms'])")
  echo "$t flow $f cvae $c"
done; done
