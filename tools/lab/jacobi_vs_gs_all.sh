O=gpurun_out/lab; mkdir -p $O
for cfg in "640 480 6 4" "640 480 6 4 --hard" "1920 1080 24 10" "1920 1080 24 10 --hard" "6200 4130 12 10"; do
  tag=$(echo $cfg | sed 's/ --hard/_hard/; s/^\([0-9]*\) \([0-9]*\) \([0-9]*\) \([0-9]*\)/\1x\2_\3v_\4src/')
  python tools/jacobi_vs_gs.py $cfg --json $O/jvg_$tag.json > $O/jacobi_vs_gs_$tag.txt 2>&1; echo "== $tag"; tail -n 22 $O/jacobi_vs_gs_$tag.txt
done
