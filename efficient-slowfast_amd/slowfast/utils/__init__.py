from slowfast._overlay import chain_package as _chain_package

_chain_package(globals())  # modules this repo does not carry resolve to the reference's slowfast/utils/
