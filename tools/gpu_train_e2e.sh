# End-to-end training rate from PNG triplets on disk (input stage N2 + train step), 832x256, bs=8, 8 decode workers.
cd $GRAFT_REPO_ROOT
D=/tmp/unflow_e2e; rm -rf $D; mkdir -p $D/data_s1/seq
python - <<PY
import numpy as np, os, sys
sys.path.insert(0, '.')
from unopticalflow_amd.evaluation import write_png
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:1125, 0:1242]
names = []
for i in range(48):
    img = np.stack([127 + 90 * np.sin(xx / (17.0 + c + i % 5) + i) * np.cos(yy / (23.0 + c)) for c in range(3)], -1)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    write_png('$D/data_s1/seq/%d.png' % i, img)
    names.append('seq/%d.png seq/%d_cam.txt' % (i, i))
open('$D/data_s1/train.txt', 'w').write('\n'.join(names) + '\n')
PY
sed "s#prepared_base_dir:.*#prepared_base_dir: '$D'#; s#num_iterations:.*#num_iterations: ${ITERS:-120}#" unopticalflow_amd/config/kitti.yaml > $D/cfg.yaml
for HOSTIN in 0 1; do
  echo "host_input=$HOSTIN"
  python -m unopticalflow_amd.train -c $D/cfg.yaml --gpu 0 --model_dir $D/models$HOSTIN --batch_size 8 --num_workers 8 \
      --log_interval 20 --save_interval 100000 --host_input $HOSTIN ${EXTRA:-} 2>&1 | grep "^iter" | tail -4
done
rm -rf $D
