for m in 64 128; do
  for D in 128 200; do for rate in 1 2; do
    timeout 300 python tools/group_sweep.py --decimations $D --rate $rate --clients 256,1024,4096 --groups 8 --modes optimized --blocks 160 --m $m 2>&1 | grep optimized | sed "s/^/streamed D$D rate$rate /"
  done; done
  timeout 300 python tools/group_sweep.py --shape config5 --clients 1024,4096 --groups 8 --modes optimized --blocks 320 --m $m --opt mix_kernel=3 2>&1 | grep optimized | sed "s/^/config5-f32mix /"
  for rate in 4 5; do
    timeout 300 python tools/group_sweep.py --decimations 100 --rate $rate --clients 1024,4096 --groups 8 --modes optimized --blocks 320 --m $m 2>&1 | grep optimized | sed "s/^/cu8-D100-rate$rate /"
  done
done
