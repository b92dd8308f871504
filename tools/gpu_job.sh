set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python tools/make_toy_audioset.py /tmp/toy --clips 1024 --val-clips 128 > gpurun_out/e2e.log 2>&1
timeout 600 python -m aum.train --model_type base --n_class 527 --label-csv /tmp/toy/class_labels_indices.csv --data-train /tmp/toy/train.json --data-val /tmp/toy/val.json -b 64 --num-workers 16 --lr 1e-4 --n-epochs 2 --freqm 48 --timem 192 --mixup 0.5 --warmup True --n-print-steps 4 --exp-dir /tmp/exp >> gpurun_out/e2e.log 2>&1; echo "train rc=$?" >> gpurun_out/e2e.log
grep -E "clips/s|mAP|rc=" gpurun_out/e2e.log | tail -12
python bench.py > gpurun_out/bench7.json 2> gpurun_out/bench7.err; tail -1 gpurun_out/bench7.json
