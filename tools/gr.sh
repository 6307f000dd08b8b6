#!/bin/bash
# development wrapper for gpurun commands: makes the scratch directory, then runs the arguments as a shell command
mkdir -p gpurun_out/r03
eval "$@"
