O=gpurun_out/r5e; mkdir -p $O
(time python -m pytest tests/test_gpu_dropin_binary.py tests/test_gpu_properties.py tests/test_gpu_errors.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
python tools/jacobi_vs_gs.py 640 480 6 4 --json $O/jvg_small.json > $O/jacobi_vs_gs_640x480_6v_4src.txt 2>&1; tail -n 3 $O/jacobi_vs_gs_640x480_6v_4src.txt
python tools/jacobi_vs_gs.py 640 480 6 4 --hard --json $O/jvg_small_hard.json > $O/jacobi_vs_gs_640x480_6v_4src_hard.txt 2>&1; tail -n 3 $O/jacobi_vs_gs_640x480_6v_4src_hard.txt
python tools/jacobi_vs_gs.py 1920 1080 24 10 --json $O/jvg_tt24.json > $O/jacobi_vs_gs_1920x1080_24v_10src.txt 2>&1; tail -n 3 $O/jacobi_vs_gs_1920x1080_24v_10src.txt
python tools/jacobi_vs_gs.py 1920 1080 24 10 --hard --json $O/jvg_tt24_hard.json > $O/jacobi_vs_gs_1920x1080_24v_10src_hard.txt 2>&1; tail -n 3 $O/jacobi_vs_gs_1920x1080_24v_10src_hard.txt
python tools/jacobi_vs_gs.py 6200 4130 12 10 --json $O/jvg_eth12.json > $O/jacobi_vs_gs_6200x4130_12v_10src.txt 2>&1; tail -n 3 $O/jacobi_vs_gs_6200x4130_12v_10src.txt
