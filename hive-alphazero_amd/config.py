"""Constants of the reference's hive_engine/config.py:8-34 and settings.py:3-4 (same names/values)."""
import string

MAX_MAP_HAFT = 6
MAX_MAP_FULL = MAX_MAP_HAFT * 2
ACTION_SPACE = MAX_MAP_FULL * MAX_MAP_FULL * 11
STATE_FEATURES = 56
MAX_GAME_LENGTH = 55
MAX_LEN_BACK = 5
SEARCH_THREADS = 32
MAX_PROCESS = 60
BOT_WEIGHT = 0.24
LOSS_WEIGHT = {"value": 1.0, "policy": 1.0}
DISCOUNTED_REWARD = 0.99

index_number = [str(i) for i in range(1, 27)][12 - MAX_MAP_HAFT:12 + MAX_MAP_HAFT]      # '7'..'18'
index_char = list(string.ascii_uppercase)[13 - MAX_MAP_HAFT:13 + MAX_MAP_HAFT]          # 'H'..'S'

PIECE_WHITE = (250, 250, 250)
PIECE_BLACK = (71, 71, 71)

# inventory_frame.py:47-99 creation order; keys as env_hive.py:76 builds them (str(type) + index)
SLOT_KEYS = (["<class 'pieces.Queen'>0"] + ["<class 'pieces.Beetle'>%d" % i for i in range(2)]
             + ["<class 'pieces.Spider'>%d" % i for i in range(2)]
             + ["<class 'pieces.Grasshopper'>%d" % i for i in range(3)]
             + ["<class 'pieces.Ant'>%d" % i for i in range(3)])
PIECE_KEYS = ["Q0", "B0", "B1", "S0", "S1", "G0", "G1", "G2", "A0", "A1", "A2"]          # env_hive.py:19-23,79
