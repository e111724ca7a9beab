"""Model constants mirrored from the reference (omchat/constants.py:7-12, omchat/make_context.py:79-80)."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
IM_START_ID = 151644
IM_END_ID = 151645
