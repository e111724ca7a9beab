"""Drop-in for the reference's single_inference.py (same flags, same call sequence tile -> prompt -> ids -> generate),
running on the HIP engine.  `python -m omchat_amd.single_inference --model-path DIR --image-path IMG --question Q`."""
import argparse
import torch

from .model.builder import load_pretrained_model
from .mm_utils import get_model_name_from_path
from .make_context import get_context


def load_image(image_file):
    from PIL import Image
    if image_file.startswith("http"):
        import requests
        from io import BytesIO
        return Image.open(BytesIO(requests.get(image_file).content)).convert("RGB")
    return Image.open(image_file).convert("RGB")


def main(args):
    from transformers import TextStreamer
    model_name = get_model_name_from_path(args.model_path)
    tokenizer, model, image_processor, context_len = load_pretrained_model(model_path=args.model_path, model_name=model_name)
    image = load_image(args.image_path) if args.image_path else None
    _, context_tokens, image_tensor = get_context(text=args.question, image=image, image_processor=image_processor,
                                                  image_grid_pinpoints=model.config.image_grid_pinpoints, tokenizer=tokenizer)
    input_ids = torch.tensor([context_tokens])
    streamer = TextStreamer(tokenizer, skip_prompt=True, skip_special_tokens=True)
    model.generation_config.pad_token_id = tokenizer.pad_token_id
    with torch.inference_mode():
        output_ids = model.generate(input_ids, images=image_tensor, do_sample=False, temperature=args.temperature, max_new_tokens=args.max_new_tokens,
                                    streamer=streamer, use_cache=True, eos_token_id=151645)
    return tokenizer.decode(output_ids[0, input_ids.shape[1]:]).strip()


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--model-path", type=str, required=True)
    p.add_argument("--image-path", type=str, default="images/extreme_ironing.jpg")
    p.add_argument("--question", type=str, default="What is unusual about this image? can you explain this to a 5-year-old kid?")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--device", type=str, choices=["cuda", "cpu"], default="cuda")
    p.add_argument("--temperature", type=float, default=0)
    p.add_argument("--max-new-tokens", type=int, default=1024)
    p.add_argument("--debug", action="store_true")
    main(p.parse_args())
