"""fthmc_amd -- MI355X-native (gfx950) hot path of field-transformation HMC for
2D U(1) lattice gauge theory, behind the Python API of nftqcd/fthmc."""
__version__ = '0.1.0'
