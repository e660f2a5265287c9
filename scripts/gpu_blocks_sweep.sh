# sub-batches on two streams share the chip: how many persistent workgroups should EACH launch have?  (one launch per step: exactly the resident capacity)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { timeout 300 python bench.py --workload $1 --max-blocks $2 --steps $3 --warmup 20 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1 blocks $2 steps $3: %.2f us per step, regions %s, frac %.3f' % (j['ms_per_step']*1e3, j['config']['region_ms_per_step'], j['roofline']['frac']))"; }
for b in 0 4096 4352 4608; do run pursuit $b 500; done
for b in 0 4352; do run pursuit $b 20; done
for b in 0 1536 1792 2048; do run pursuit_c5 $b 200; done
for b in 0 3072 3584 4096; do run pursuit_colocate $b 200; done
