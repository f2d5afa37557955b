for N in 5000 20000 100000 500000; do for P in i8x3 f16x2 i8f16 f16; do
python bench.py --steps 1 --warmup 0 --n-indv $N --n-snp 200000 --jn 100 --precision $P --no-cpu-baseline --no-alt --verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$N $P', repr(c['sigma_total']), repr(c['sigma_se']))"
done; done
