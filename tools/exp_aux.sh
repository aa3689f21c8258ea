for a in 0 1; do for b in 128 192 256 512 1024; do echo "AUX=$a BLOCKS=$b"; NVSF_SL_AUX=$a NVSF_SL_BLOCKS=$b python tools/ab_density.py S 2>&1 | grep median; done; done
