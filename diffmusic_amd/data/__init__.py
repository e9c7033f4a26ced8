from .dataloader import MP3Dataset, WAVDataset, get_dataloader, get_dataset, register_dataset   # noqa: F401
