# A/B of the in-tile decode on the configs[4] stream, both codecs, one process each (scripts/bench_hybrid_stream.py)
mkdir -p gpurun_out/r05z
for codec in freqs_only full; do
  CODEC=$codec MODES=warm,cold CYCLES=3 CONFIGS="in_tile_decode:;decode_kernel:hybrid_cold_fused=0" OUT=r05z/cold_ab_$codec.json timeout 600 python scripts/bench_hybrid_stream.py > gpurun_out/r05z/cold_ab_$codec.log 2>&1
  tail -3 gpurun_out/r05z/cold_ab_$codec.log | cut -c1-900
done
