from .build import build
print(build(verbose=True))
