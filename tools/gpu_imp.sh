cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python tools/profile_importance.py 8 2>&1 | tail -2
python tools/profile_importance.py 8 1 2>&1 | tail -2 | sed "s/^/sb=1 /"
python tools/time_training_shapes.py 20 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_reference_config.py -m gpu -x -q 2>&1 | tail -3
