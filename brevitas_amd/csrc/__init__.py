"""HIP sources of libbvq.so and the script that builds them (python -m brevitas_amd.csrc.build)."""
