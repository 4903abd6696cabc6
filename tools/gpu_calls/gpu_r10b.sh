for v in "" "EXL2_ATT_RB=4"; do
echo "== seed 35 $v"
env $v EXL2_TEST_SEEDS=36 timeout 600 python -m pytest tests/test_chain.py -m gpu -q -x -k "test_chain_decode_random_models and hip-35" 2>&1 | grep -E "^E  |passed|failed|assert" | head -12
done
