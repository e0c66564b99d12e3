"""WIDER FACE evaluation with the reference's module and method names (/root/reference/eval/), matching and PR accumulation on the GPU."""
