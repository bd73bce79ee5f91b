for fpl in 8 16 32; do
  EMULATE=8 FPL=$fpl BATCH=$fpl TILE=3,3 FLIGHTS=2,3 SKIP=copy timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
done
EMULATE=4 FPL=32 BATCH=32 TILE=3,3 FLIGHTS=2 SKIP=copy timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
EMULATE=2 FPL=16 BATCH=16 TILE=3,3 FLIGHTS=2 SKIP=copy timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
