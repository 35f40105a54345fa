# k_conv_dec_nchw compile-time variants (scripts/build_variant.sh <suffix> qcnn_decoded.hip -D...): per-layer times at 1000 and 125 images
for v in "" ${VARIANTS:-_w12 _w16}; do
echo "variant [$v]"
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$v.so python scripts/layer_times.py 1000 10 1 | grep -E "_conv" | cut -c1-40
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$v.so python scripts/layer_times.py 125 30 1 | grep -E "_conv" | cut -c1-40
done
