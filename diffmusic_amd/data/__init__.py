from .dataloader import WAVDataset, get_dataloader, get_dataset, register_dataset   # noqa: F401
