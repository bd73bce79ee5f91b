for fpl in 8 4; do
  EMULATE=8 FPL=$fpl BATCH=$fpl TILE=3,3 FLIGHTS=3,4,6,8 SKIP=copy timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
done
EMULATE=8 FPL=8 BATCH=8 TILE=2,3 FLIGHTS=3,6 SKIP=copy timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
EMULATE=8 FPL=8 BATCH=8 TILE=2,2 FLIGHTS=3,6 SKIP=copy timeout -k 10 200 python tools/exchange_host_cost.py 2>&1 | grep "host enqueue"
