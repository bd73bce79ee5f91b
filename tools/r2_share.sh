# one rank's share of an N-rank job on one GPU (projection only; same camera for every frame)
for n in 1 2 4 8; do
  EMULATE=$n FPL=8 BATCH=8 TILE=3,3 FLIGHTS=2 SKIP=gather,deint timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
done
for n in 2 8; do
  EMULATE=$n FPL=8 BATCH=8 TILE=3,3 FLIGHTS=2 SKIP=copy timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
done
