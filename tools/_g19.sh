O=gpurun_out/r03q; mkdir -p $O
export NGSAMG_NO_BUILD=1
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -k "elasticity_block_hybrid or block_hybrid_gs or elasticity_hybrid_gs" > $O/distb.log 2>&1; echo "rc=$?" >> $O/distb.log
tail -40 $O/distb.log
