for m in 0 3; do echo "GLIO_KNN_MODE=$m"; GLIO_KNN_MODE=$m timeout 200 python scripts/stream_cpp_ab.py 2>&1 | head -2; done
